// spamtree_hip.hip -- gfx950 kernels + C-ABI for spamtree's per-Gibbs-sweep DAG-node linear algebra.
//
// What each kernel replaces in the reference (paths relative to /root/reference/src):
//   k_factor<MODE_FACTOR>  spamtree_model.cpp:834-998  get_loglik_comps_w_std  (phase A)  + covariance_functions.cpp:95-111, 213-286
//   k_factor<MODE_PREDICT> spamtree_model.cpp:1234-1358 predict_std            (phase P)
//   k_sample               spamtree_model.cpp:1011-1226 gibbs_sample_w_std     (phase B)
//   k_loglik               spamtree_model.cpp:781-826  get_loglik_w_std        (phase C)
//   k_stats / k_xb         spamtree_model.cpp:1374-1375, 1382, 1397-1400       (beta / tausq sufficient statistics)
//
// Design (DESIGN.md has the derivation): the reference materialises, per block u with ancestor rows PI_u,
// H_u = K_{u,pa} K_{pa,pa}^{-1}, the dense (P+m)^2 inverse Cholesky of K_{[pa,u]} and its Gram matrix.  The inverse
// Cholesky of an ancestor set is block lower triangular and its row panel for block a is [-Ri_a H_a | Ri_a]
// (tree_utils.cpp:204-206), so this build stores exactly ONE array per block, that panel ("Linv panel",
// m x (P+m), row-major), and every phase works from the chain of ancestor panels:
//   V = Linv_pa * K_{pa,u}            (one pass over the chain, ancestors last-to-first so V overwrites K in place)
//   T = V' * Linv_pa = H_u            (accumulated in the same pass)
//   R = K_uu - V'V,  Ri = chol(R)^{-1},  panel_u = [-Ri*T | Ri]
// Messages to ancestors are pushed as per-ancestor (m_a x m_a, m_a) pairs and summed hierarchically through
// direct children in a fixed order (no FP64 atomics -> bit-reproducible for any launch geometry).
//
// This translation unit is the HOST side: handle, C-ABI, launches.  The kernels live in one translation unit per family
// (k_factor_generic.hip, k_factor_mfma.hip, k_factor_quad.hip, k_factor_wide.hip, k_sample.hip, k_misc.hip); the headers
// included here give their argument structures, launch constants and prototypes.

#include "st_device.hpp"
#include "factor_generic.hpp"
#include "factor_mfma.hpp"
#include "chol_blocked.hpp"
#include "factor_quad.hpp"
#include "factor_big.hpp"
#include "factor_wide.hpp"
#include "factor_lchain.hpp"
#include "sample_kernels.hpp"
#include "misc_kernels.hpp"

// ===============================================================================================================
// host side
// ===============================================================================================================
static thread_local std::string g_create_error;

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    n = count;
    if (count == 0) { p = nullptr; return hipSuccess; }
    return hipMalloc((void **)&p, count * sizeof(T));
  }
  hipError_t upload(const std::vector<T> &v) {
    hipError_t e = alloc(v.size());
    if (e != hipSuccess || v.empty()) return e;
    return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  }
  void free() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

struct LevelInfo {
  int first = 0, count = 0;   // into lvl_list
  int isref = 1;
  int maxP = 0, maxM = 0, maxMa = 0, maxLd = 0;
  bool big_factor = false, big_sample = false, sample_sq = false;   // sample_sq: k_sample<true> keeps S and chol(S)^-1 in LDS
  size_t lds_factor = 0, lds_sample = 0, lds_loglik = 0;
  double alg_bytes_A = 0, alg_bytes_B = 0, alg_bytes_C = 0, alg_bytes_msg = 0;
  double flops_A = 0, flops_B = 0, flops_C = 0;
  // MFMA fast path of phase A (column groups)
  bool fast = false;
  int grp_first = 0, grp_count = 0, Pm4 = 0, ldKV = 2, ldS = 2, SRm = 1, stage_dbl = 0;
  size_t lds_fast = 0;
  int ldN = 2, Mr4 = 4, Mrows = 1, av_dbl = 224, maxJ = 0;
  size_t lds_sfast = 0, lds_slean = 0;
  bool bigmfma = false;            // generic level whose phase A takes k_factor_bigmfma
  int wide_first = 0, wide_count = 0, wide_maxN = 0;   // sibling groups of this rank's run (k_factor_wide); count 0: not used
  size_t lds_wide = 0;
  int bm_ldS = 0;
  size_t lds_bigmfma = 0;
  int lchain = 0;                  // non-reference level on k_factor_lchain<lchain> (0: not used)
  bool lchain_ref = false;         // ... a REFERENCE level: k_factor_lchain, then k_factor_ref_finish
  size_t lds_ref_finish = 0;
  int rf_first = 0;                // its blocks' entries (this rank's run) in d_rfvoff / d_rfvld
  int lc_first = 0, lc_count = 0;  // its slabs (this rank's run) in d_lcslabs
  int quad_first = 0, quad_count = 0, qown_lo = 0, qown_n = 0, q_ldS = 0, q_nkx = 0;   // k_factor_quad (q_nkx = 0: not eligible)
  size_t lds_quad = 0;
  int own_lo = 0, own_n = 0, gown_lo = 0, gown_n = 0;   // this rank's run of the level's block list / group list
};

struct st_handle_s {
  std::string err;
  int device = 0;
  hipStream_t stream = nullptr;
  int quirks = 1, force_generic = 0;
  long long n_all = 0, n_blocks = 0;
  int q = 1, p = 1, d = 2, n_groups = 0, n_actual_groups = 0;
  long long n_obs = 0;
  size_t lds_limit = 65536;
  int sm_count = 256;

  std::vector<long long> dev2model, model2dev;       // rows
  std::vector<int> blk_model2dev;                    // blocks
  std::vector<Blk> blks;                             // device block order
  std::vector<int> anc_idx, dch_idx, lvl_list, pred_list, all_obs_list;
  std::vector<Grp> grps;
  DevBuf<Grp> d_grps;
  std::vector<Quad> quads;
  DevBuf<Quad> d_quads;
  std::vector<WideGrp> wgrps;                 // sibling groups of the wide levels (k_factor_wide)
  DevBuf<WideGrp> d_wgrps;
  int wide_on = 1;                            // SPAMTREE_WIDE=0: k_factor_bigmfma (one block per workgroup) instead
  std::vector<LcSlab> lcslabs;                // k_factor_lchain: slabs of sibling groups
  DevBuf<LcSlab> d_lcslabs;
  std::vector<long long> rfvoff;   // k_factor_ref_finish: per block of a reference level on the lchain route, its columns in the V scratch
  DevBuf<long long> d_rfvoff;
  DevBuf<double> d_vscr;                      // V = Linv_pa K_pa,u of ONE such level (the largest): written by k_factor_lchain, read by k_factor_ref_finish
  size_t vscr_need = 0;
  DevBuf<double> d_lcrow;                     // per-row e^2 | log r of the lchain levels (2 n)
  DevBuf<double> d_s0;                        // Ri' Ri of the reference blocks on the generic phase-B path (theta-only, cached with the Gram parts)
  DevBuf<long long> d_s0off;                  // per block: offset into d_s0, -1 = none
  bool c_pending = false;                     // st_sample_w_loglik_begin: the sweep's failure word and log-density are on their way to pin[8..10]
  int c_rc = 0; double c_ll = 0.0;            // ... or (multi-GPU / communicator attached) already here
  int gram_big = 1;                           // SPAMTREE_GRAM_BIG=0: the generic sweep kernel rebuilds the records' Gram parts itself (one thread per entry)
  int lchain_on = 1;                          // SPAMTREE_LCHAIN=0: non-reference long-chain levels stay on k_factor_wide / k_factor_bigmfma
  std::vector<long long> gdesc;               // group descriptors (GdHead layout), gd_stride words per group
  DevBuf<long long> d_gdesc;
  int gd_stride = 8;
  int quad_nu = 4;
  // multi-GPU sharding
  int rank = 0, world = 1, cut = 0;
  std::vector<int> blk_owner;                 // device block -> owning rank, -1 = replicated
  std::vector<int> own_obs_list;              // observed blocks this rank evaluates in phase C
  DevBuf<int> d_ownobs;
  std::vector<int> own_grp_list, own_obs_slow; // the same set split: column groups of the fast levels / blocks of the others
  DevBuf<int> d_owngrp, d_ownslow;
  DevBuf<unsigned char> d_rowmask, d_blkmask; // 1 = this rank contributes the entry to a sum-with-zeros exchange
  DevBuf<double> d_comm;                      // 2*n_blocks + 64 doubles
  DevBuf<double> d_gather;                    // all-gather of w: world x gather_cnt (a rank's owned rows in device order + its failure word)
  DevBuf<int> d_gidx;                         // device row of every slot of d_gather (-1: padding / the failure word)
  int gather_cnt = 1;
  DevBuf<double> d_gerr;                      // the ranks' failure words after the all-gather (64)
  // phase A of the latency-bound top levels ahead of time (st_factor_begin): they depend on theta only -- except for the
  // blocks' quadratic forms, redone with the current w afterwards -- and run on a second stream under the sweep
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_top = nullptr, ev_main = nullptr;
  DevBuf<int> d_err2, d_toplist;
  int n_toplist = 0, g_top = 0;
  bool async_top = false, top_pending = false, prof_suspend = false, async_top_off = false;
  hipEvent_t ev_stats = nullptr;
  bool stats_on_stream2 = false;   // the statistics kernels of the current (w, XB) are in flight on the second stream
  bool stats_prefetched = false;   // ... and their results follow them to pin[20 ..] on that stream
  int top_phys = -1;
  std::vector<double> top_theta;
  long long top_off = 0, top_len = 0;         // message records of the cut level inside `acc`
  std::vector<std::pair<long long, long long>> top_zero;   // sub-ranges of it owned by other ranks
  bool ext_stream = false;
  DevBuf<double> d_sum_w, d_sum_yhat;         // running sums over saved iterations (st_summary_*)
  long long n_summary = 0;
  DevBuf<double> d_draws_w, d_draws_yhat;     // st_summary_reserve: the saved draws themselves, [keep][n_all] (quantiles)
  long long draws_cap = 0, n_draws = 0;
  int factor_gen = 1;
  int sample_lean = 1;                        // sweeps with cached Gram parts take k_sample_lean (SPAMTREE_SAMPLE_LEAN=0: never)
  int sample_wave = 1;                        // reference blocks of <= 27 rows: one block per wave (SPAMTREE_SAMPLE_WAVE=0: k_sample_lean)
  int lchain_ref_on = 1, lchain_ref_min = 1;   // reference levels of wide-block trees (with at least that many blocks): k_factor_lchain + k_factor_ref_finish
  int leaf_wide = 1;                          // k_sample_leaf_wide for the non-reference levels of wide-block trees (SPAMTREE_LEAF_WIDE=0: the generic kernel)
  int leaf_seg = 1;                           // k_sample_leaf_seg (segment-aligned lanes) where eligible
  int gram_direct_level = -1;                 // >= 0: that (last reference) level forms its children's Gram parts itself: k_gram_direct
  int split_gram = 1;                         // sweeps that rebuild the Gram parts: k_gram + lean kernels (SPAMTREE_SPLIT_GRAM=0: k_sample_mfma)
  bool stats_valid = false;                   // d_stats matches the current w and XB
  bool host_stats_valid = false;              // ... and host_stats holds a copy of it
  std::vector<double> host_stats;
  double *pin = nullptr;                      // 64 doubles of pinned host memory for the small device-to-host reads: [0..3] st_factor (comm path) /
                                              // st_loglik_w sums + failure word, [8..11] st_sample_w_loglik_end, [12..15] st_factor_enqueue / _finish, [20..] statistics
  double *pin_up = nullptr;                   // pinned staging of the small per-iteration uploads (beta, tausq_inv): two slots taken in turn,
  int pin_up_slot = 0, pin_up_len = 0;        // so that the copy is truly asynchronous and the setters need no host synchronisation
  hipEvent_t ev_up[2] = {nullptr, nullptr};   // recorded behind a slot's copy: a slot is rewritten only after its last copy has run
  bool factor_open = false; int factor_open_slot = 0;   // st_factor_enqueue without its st_factor_finish yet
  std::vector<double> top_theta_open;                   // ... its theta where the work itself waits for st_factor_finish (communicator attached)
  hipEvent_t ev_factor = nullptr;                       // behind the copies of an enqueued factorisation's sums and failure word
  std::vector<char> s0_valid;                 // per level: d_s0 holds the theta-only precision parts of the accepted theta (column-group levels)
  bool gram_valid = false;                    // message Gram parts in `acc` match the accepted theta (slot 0)
  bool cache_gram = true;
  bool limited = false;               // limited_tree: single parents, marginal chain factors (k_marginal_invchol)
  std::vector<int> twin_list;         // limited_tree: device ids of the blocks that own a chain panel
  DevBuf<int> d_twin;
  int twin_maxM = 1;
  ncclComm_t comm = nullptr;                  // native RCCL communicator (st_comm_init); null = exchanges are the caller's
  std::vector<LevelInfo> levels;
  LevelInfo pred_info;
  int pred_grp_first = 0, pred_grp_count = 0, pred_quad_first = 0, pred_quad_count = 0, pred_nkx = 0;   // phase P on k_factor_quad's leaf path (pred_nkx = 0: generic kernel)
  size_t pred_lds = 0;
  std::vector<double> xtx;
  std::vector<long long> n_obs_q;

  DevBuf<double> d_cx, d_cy, d_y, d_X, d_w, d_xb, d_z, d_B, d_panels[2], d_acc, d_logdet[2], d_loglik[2], d_scalars, d_partial,
      d_stats, d_scratch, d_tmp_n, d_tsq;
  DevBuf<int> d_mv, d_anc, d_dch, d_lvl, d_pred, d_allobs, d_err;
  DevBuf<unsigned char> d_obs;
  DevBuf<long long> d_dev2model, d_partner;
  DevBuf<Blk> d_blks;
  size_t panel_total = 0, acc_total = 0;
  long long scratch_stride = 0;
  int scratch_wgs = 0;
  int slot_map[2] = {0, 1};    // logical slot (0 param, 1 alter) -> physical arena
  double tausq_inv[QMAX];
  std::vector<double> theta[2];
  bool z_valid = false;

  // profiling
  int prof = 0;   // 0 off, 1 every kernel family, 2 phase A only (the roofline measurement at the lowest cost)
  double prof_ms[ST_N_KERNEL_FAMILIES] = {0};
  long long prof_n[ST_N_KERNEL_FAMILIES] = {0};
  std::vector<double> prof_level_ms;   // phase-A time per level, accumulated
  std::vector<long long> prof_level_n;
  struct ProfRec { hipEvent_t a, b; int fam, level, count; };   // count: kernel launches inside the bracket
  std::vector<ProfRec> prof_pending;
  std::vector<hipEvent_t> ev_free;
};

#define HCHK(h, call)                                                                                         \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) {                                                                                   \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                           \
      return ST_ERR_HIP;                                                                                      \
    }                                                                                                         \
  } while (0)

static int fail_create(st_handle_s *h, int code, const std::string &msg) {
  g_create_error = msg;
  if (h) {
    st_destroy(h);
  }
  return code;
}

// Launch timing with HIP events on the launch stream, harvested lazily (no host sync inside the measured region).
static hipEvent_t prof_event(st_handle_s *h) {
  if (!h->ev_free.empty()) { hipEvent_t e = h->ev_free.back(); h->ev_free.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
static void prof_harvest(st_handle_s *h) {
  for (auto &r : h->prof_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      h->prof_ms[r.fam] += ms;
      h->prof_n[r.fam] += r.count;
      if (r.level >= 0 && r.level < (int)h->prof_level_ms.size()) { h->prof_level_ms[r.level] += ms; h->prof_level_n[r.level] += 1; }
    }
    h->ev_free.push_back(r.a);
    h->ev_free.push_back(r.b);
  }
  h->prof_pending.clear();
}
struct ProfScope {
  st_handle_s *h;
  st_handle_s::ProfRec r;
  // mode 1: every launch is bracketed; mode 2: only the whole-phase bracket of phase A (level == -2), one pair of events
  hipStream_t st;
  ProfScope(st_handle_s *h_, int fam, int level = -1, int count = 1, hipStream_t st_ = nullptr) : h(h_), st(st_ ? st_ : h_->stream) {
    r.fam = fam; r.level = level; r.count = count; r.a = r.b = nullptr;
    const bool on = level == -2 ? (h->prof == 2 && !h->prof_suspend) : h->prof == 1;
    if (on) { r.a = prof_event(h); r.b = prof_event(h); (void)hipEventRecord(r.a, st); }
  }
  ~ProfScope() {
    if (r.a && r.b) {
      (void)hipEventRecord(r.b, st);
      h->prof_pending.push_back(r);
      if (h->prof_pending.size() > 8192) prof_harvest(h);
    }
  }
};

// w or XB is about to change: the cached statistics die; a reduction still in flight on the second stream finishes first
static void invalidate_stats(st_handle_s *h) {
  if (h->stats_on_stream2) {
    (void)hipSetDevice(h->device);
    (void)hipStreamWaitEvent(h->stream, h->ev_stats, 0);
    h->stats_on_stream2 = false;
  }
  h->stats_valid = false; h->host_stats_valid = false;
}

static size_t lds_factor_bytes(int maxP, int maxM, int maxMa, int SR, bool big) {
  size_t dbl = (size_t)3 * (maxP + maxM) + 3 * (size_t)maxM + (size_t)SR * maxP;
  size_t bytes = dbl * 8 + (size_t)((maxP + maxM + 1) & ~1) * 4;
  if (!big) bytes += ((size_t)2 * maxP * maxM + (size_t)maxMa * maxM + (size_t)2 * maxM * maxM) * 8;
  return bytes + 64;
}
static size_t scratch_factor_doubles(int maxP, int maxM, int maxMa) {
  return (size_t)2 * maxP * maxM + (size_t)maxMa * maxM + (size_t)2 * maxM * maxM;
}
static size_t lds_sample_sq_bytes(int maxM) { return ((size_t)maxM * ((maxM + 7) | 1) + maxM + 16) * 8; }   // S (odd row stride) + the pivot column
static size_t lds_sample_bytes(int maxP, int maxM, int maxLd, bool big) {
  size_t dbl = (size_t)(maxP + maxM) + 4 * (size_t)maxM + (size_t)MAXJ * maxM;   // ... + segment sums seg[t][r]
  if (!big) dbl += (size_t)maxM * maxLd + (size_t)maxM * maxM;
  return dbl * 8 + 64;
}
static size_t lds_loglik_bytes(int maxP, int maxM) { return ((size_t)maxP + 2 * (size_t)maxM) * 8 + 64; }

extern "C" const char *st_last_error(st_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }
extern "C" void st_set_create_error(const char *msg) { g_create_error = msg ? msg : ""; }   // other translation units of the library

extern "C" int st_destroy(st_handle h) {
  if (!h) return ST_OK;
  (void)hipSetDevice(h->device);
  h->d_cx.free(); h->d_cy.free(); h->d_y.free(); h->d_X.free(); h->d_w.free(); h->d_xb.free(); h->d_z.free(); h->d_B.free();
  h->d_panels[0].free(); h->d_panels[1].free(); h->d_acc.free();
  for (int s = 0; s < 2; ++s) { h->d_logdet[s].free(); h->d_loglik[s].free(); }
  h->d_scalars.free(); h->d_partial.free(); h->d_stats.free(); h->d_scratch.free(); h->d_tmp_n.free(); h->d_tsq.free();
  h->d_mv.free(); h->d_anc.free(); h->d_dch.free(); h->d_lvl.free(); h->d_pred.free(); h->d_allobs.free(); h->d_err.free();
  h->d_twin.free(); h->d_wgrps.free(); h->d_lcslabs.free(); h->d_lcrow.free(); h->d_rfvoff.free(); h->d_vscr.free(); h->d_s0.free(); h->d_s0off.free(); h->d_obs.free(); h->d_dev2model.free(); h->d_partner.free(); h->d_blks.free(); h->d_grps.free(); h->d_quads.free(); h->d_gdesc.free();
  h->d_ownobs.free(); h->d_owngrp.free(); h->d_ownslow.free(); h->d_rowmask.free(); h->d_blkmask.free(); h->d_comm.free(); h->d_gather.free(); h->d_gidx.free(); h->d_gerr.free(); h->d_err2.free(); h->d_toplist.free();
  if (h->ev_top) (void)hipEventDestroy(h->ev_top);
  if (h->ev_main) (void)hipEventDestroy(h->ev_main);
  if (h->ev_stats) (void)hipEventDestroy(h->ev_stats);
  if (h->stream2) (void)hipStreamDestroy(h->stream2); h->d_sum_w.free(); h->d_sum_yhat.free(); h->d_draws_w.free(); h->d_draws_yhat.free();
  prof_harvest(h);
  for (auto e : h->ev_free) (void)hipEventDestroy(e);
  if (h->comm) (void)ncclCommDestroy(h->comm);
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->pin_up) (void)hipHostFree(h->pin_up);
  for (int i = 0; i < 2; ++i) if (h->ev_up[i]) (void)hipEventDestroy(h->ev_up[i]);
  if (h->ev_factor) (void)hipEventDestroy(h->ev_factor);
  if (h->stream && !h->ext_stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return ST_OK;
}

static int create_impl(const st_problem *pb, const st_options *opt, st_handle *out, bool plan_only, int64_t *owner_out, int32_t *cut_out);

extern "C" int st_create(const st_problem *pb, const st_options *opt, st_handle *out) {
  return create_impl(pb, opt, out, false, nullptr, nullptr);
}

// Pure host: which rank owns each block (0-based block id) when `world` processes share one problem; -1 = replicated
// on every rank (the levels above the cut).  No GPU needed: lets the sharding plan be tested on CPU.
extern "C" int st_shard_plan(const st_problem *pb, int32_t world, int64_t *owner, int32_t *cut_level) {
  st_options opt = {0, 1, 0, world, 0, 0};
  st_handle dummy = nullptr;
  return create_impl(pb, &opt, &dummy, true, owner, cut_level);
}
// the same for a problem that needs options to be read at all (limited_tree: st_options.reserved bit 1); rank / device of `opt` are ignored
extern "C" int st_shard_plan_opt(const st_problem *pb, const st_options *opt_in, int32_t world, int64_t *owner, int32_t *cut_level) {
  st_options opt = {0, 1, 0, world, 0, 0};
  if (opt_in) { opt.reference_quirks = opt_in->reference_quirks; opt.force_generic = opt_in->force_generic; opt.reserved = opt_in->reserved; }
  st_handle dummy = nullptr;
  return create_impl(pb, &opt, &dummy, true, owner, cut_level);
}

static int create_impl(const st_problem *pb, const st_options *opt, st_handle *out, bool plan_only, int64_t *owner_out, int32_t *cut_out) {
  if (!pb || !out) { g_create_error = "st_create: null argument"; return ST_ERR_USAGE; }
  *out = nullptr;
  if (pb->d != 2) { g_create_error = "only d=2 is reachable from spamtree() (R/spamtree_fit.R:58-60)"; return ST_ERR_UNSUPPORTED; }
  if (pb->q < 1 || pb->q > QMAX) { g_create_error = "q out of range"; return ST_ERR_UNSUPPORTED; }
  if (pb->p < 1 || pb->p > 8) { g_create_error = "p must be in 1..8"; return ST_ERR_UNSUPPORTED; }
  if (opt && (opt->world < 1 || opt->rank < 0 || opt->rank >= opt->world || opt->world > 64)) { g_create_error = "bad rank/world"; return ST_ERR_USAGE; }
  st_handle_s *h = new st_handle_s();
  h->rank = opt ? opt->rank : 0;
  h->world = opt ? opt->world : 1;
  h->device = opt ? opt->device : 0;
  h->quirks = opt ? opt->reference_quirks : 1;
  h->force_generic = opt ? opt->force_generic : 0;
  h->cache_gram = !(opt && (opt->reserved & 1));
  h->limited = opt && (opt->reserved & 2);
  const long long n = pb->n_all, nb = pb->n_blocks;
  h->n_all = n; h->n_blocks = nb; h->q = pb->q; h->p = pb->p; h->d = pb->d; h->n_groups = pb->n_groups;
  for (int j = 0; j < QMAX; ++j) h->tausq_inv[j] = 1.0;

  // ---- block census (na_study :303-313), levels (make_gibbs_groups :194-301)
  std::vector<int> m_of(nb), obs_of(nb, 0), grp_of(nb);
  std::vector<long long> labels(pb->block_groups, pb->block_groups + nb);
  std::sort(labels.begin(), labels.end());
  labels.erase(std::unique(labels.begin(), labels.end()), labels.end());
  if ((int)labels.size() > pb->n_groups) return fail_create(h, ST_ERR_TOPOLOGY, "more levels in block_groups than entries in res_is_ref");
  std::vector<char> row_seen(n, 0);
  for (long long u = 0; u < nb; ++u) {
    m_of[u] = (int)(pb->indexing_ptr[u + 1] - pb->indexing_ptr[u]);
    grp_of[u] = (int)(std::lower_bound(labels.begin(), labels.end(), pb->block_groups[u]) - labels.begin());
    for (long long k = pb->indexing_ptr[u]; k < pb->indexing_ptr[u + 1]; ++k) {
      const long long r = pb->indexing_idx[k];
      if (r < 0 || r >= n || row_seen[r]) return fail_create(h, ST_ERR_TOPOLOGY, "indexing is not a partition of the rows");
      row_seen[r] = 1;
      if (std::isfinite(pb->y[r])) obs_of[u]++;
    }
  }
  for (long long r = 0; r < n; ++r)
    if (!row_seen[r]) return fail_create(h, ST_ERR_TOPOLOGY, "row without a block");
  const int G = (int)labels.size();
  std::vector<int> grp_has_obs(G, 0);
  for (long long u = 0; u < nb; ++u)
    if (obs_of[u] > 0) grp_has_obs[grp_of[u]] = 1;
  int n_actual = 0;
  for (int g = 0; g < G; ++g) n_actual += grp_has_obs[g];
  for (int g = 0; g < n_actual; ++g)
    if (!grp_has_obs[g]) return fail_create(h, ST_ERR_TOPOLOGY, "an empty level precedes an observed one");
  h->n_actual_groups = n_actual;

  // ---- device block order: by (level, id); rows contiguous per block
  std::vector<int> order(nb);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return grp_of[a] < grp_of[b]; });
  h->blk_model2dev.assign(nb, -1);
  {
    // inside a level, blocks with the same last parent (identical ancestor chain) are made contiguous, so a
    // workgroup can take several sibling leaf blocks as one column group
    int i0 = 0;
    while (i0 < nb) {
      int i1 = i0;
      while (i1 < nb && grp_of[order[i1]] == grp_of[order[i0]]) ++i1;
      auto key = [&](int u) -> long long {
        const long long p0 = pb->parents_ptr[u], p1 = pb->parents_ptr[u + 1];
        if (p1 == p0) return -1;
        const long long a = pb->parents_idx[p1 - 1];
        return (a >= 0 && a < nb) ? (long long)h->blk_model2dev[a] : -1;
      };
      std::stable_sort(order.begin() + i0, order.begin() + i1, [&](int a, int b) { return key(a) < key(b); });
      for (int i = i0; i < i1; ++i) h->blk_model2dev[order[i]] = i;
      i0 = i1;
    }
  }
  h->dev2model.resize(n); h->model2dev.resize(n);
  h->blks.resize(nb);
  long long row = 0, panel_total = 0, acc_total = 0;
  for (int i = 0; i < nb; ++i) {
    const int u = order[i];
    Blk &B = h->blks[i];
    B.row0 = row; B.m = m_of[u]; B.level = grp_of[u]; B.model_id = u; B.nobs = obs_of[u];
    for (long long k = pb->indexing_ptr[u]; k < pb->indexing_ptr[u + 1]; ++k) {
      if (k > pb->indexing_ptr[u] && pb->indexing_idx[k] <= pb->indexing_idx[k - 1])
        return fail_create(h, ST_ERR_TOPOLOGY, "indexing(u) must be ascending");
      h->dev2model[row] = pb->indexing_idx[k];
      h->model2dev[pb->indexing_idx[k]] = row;
      ++row;
    }
  }
  // ---- ancestors: chain property anc(u) = anc(last parent) + [last parent]
  for (int i = 0; i < nb; ++i) {
    const int u = order[i];
    Blk &B = h->blks[i];
    const long long p0 = pb->parents_ptr[u], p1 = pb->parents_ptr[u + 1];
    B.nanc = (int)(p1 - p0);
    if (B.nanc > MAXJ) return fail_create(h, ST_ERR_UNSUPPORTED, "more than ST_MAX_ANCESTORS ancestors");
    B.anc_ptr = (int)h->anc_idx.size();
    int P = 0;
    for (long long k = p0; k < p1; ++k) {
      const long long a = pb->parents_idx[k];
      if (a < 0 || a >= nb) return fail_create(h, ST_ERR_TOPOLOGY, "parent id out of range");
      if (k > p0 && a <= pb->parents_idx[k - 1]) return fail_create(h, ST_ERR_TOPOLOGY, "parents(u) must be ascending");
      if (grp_of[a] >= grp_of[u]) return fail_create(h, ST_ERR_TOPOLOGY, "parent on the same or a deeper level");
      if (pb->res_is_ref[grp_of[a]] != 1) return fail_create(h, ST_ERR_TOPOLOGY, "parent on a non-reference level");
      if (obs_of[a] == 0) return fail_create(h, ST_ERR_TOPOLOGY, "ancestor block without observations");
      h->anc_idx.push_back(h->blk_model2dev[a]);
      P += m_of[a];
    }
    B.P = P;
    if (h->limited) {
      if (B.nanc > 1) return fail_create(h, ST_ERR_TOPOLOGY, "limited_tree: a block has more than one parent (make_edges_limited gives one)");
    } else if (B.nanc > 0) {
      const long long last = pb->parents_idx[p1 - 1];
      const long long q0 = pb->parents_ptr[last], q1 = pb->parents_ptr[last + 1];
      bool ok = (q1 - q0) == (p1 - p0 - 1);
      for (long long k = 0; ok && k < q1 - q0; ++k) ok = pb->parents_idx[q0 + k] == pb->parents_idx[p0 + k];
      if (!ok) return fail_create(h, ST_ERR_UNSUPPORTED, "parents(u) is not parents(last parent)+[last parent]: for make_edges_limited's single-parent lists set the limited_tree bit of st_options");
    }
    const bool observed = B.nobs > 0;
    B.isref = (observed && B.level < pb->n_groups && pb->res_is_ref[B.level] == 1) ? 1 : 0;
    B.ld = B.P + (B.isref ? B.m : 1);
    B.panel_off = -1; B.acc_off = 0; B.acc_len = 0;
    B.chain_off = -1;
    if (observed) {
      B.panel_off = panel_total;
      panel_total += (long long)B.m * B.ld;
      B.chain_off = B.panel_off;
      if (h->limited) {
        B.chain_off = -1;
        if (B.isref) {   // every observed reference block may be somebody's parent (observed or prediction children)
          B.chain_off = panel_total;
          panel_total += (long long)B.m * B.m;
          h->twin_list.push_back(i);
          h->twin_maxM = std::max(h->twin_maxM, B.m);
        }
      }
    }
  }
  // acc layout + direct children
  std::vector<std::vector<int>> dch(nb);
  for (int i = 0; i < nb; ++i) {
    Blk &B = h->blks[i];
    if (B.nobs == 0) continue;
    long long len = 0;
    for (int t = 0; t < B.nanc; ++t) {
      const int ma = h->blks[h->anc_idx[B.anc_ptr + t]].m;
      len += (long long)ma * ma + ma;
    }
    if (len > INT_MAX) return fail_create(h, ST_ERR_UNSUPPORTED, "message record too large");
    B.acc_len = (int)len;
    B.acc_off = acc_total;
    acc_total += len;
    if (B.nanc > 0) dch[h->anc_idx[B.anc_ptr + B.nanc - 1]].push_back(i);
  }
  h->panel_total = (size_t)panel_total;
  h->acc_total = (size_t)acc_total;

  // ---- multi-GPU ownership: whole subtrees below a cut level go to one rank, the levels
  // above the cut are replicated.  cut = first reference level (not the last observed one) with >= 2*world
  // observed blocks; its blocks, contiguous in device order, are split into `world` runs of equal weight
  // (weight = sum over the subtree of m*P^2, the factorisation cost).
  h->blk_owner.assign(nb, -1);
  h->cut = n_actual;            // nothing sharded unless a cut is found
  if (h->world > 1) {
    std::vector<int> cnt(G, 0);
    for (int i = 0; i < nb; ++i) if (h->blks[i].nobs > 0) cnt[h->blks[i].level]++;
    for (int g = 0; g + 1 < n_actual; ++g)
      if (pb->res_is_ref[g] == 1 && cnt[g] >= 2 * h->world) { h->cut = g; break; }
    if (h->cut < n_actual) {
      const int cut = h->cut;
      std::vector<int> root_of(nb, -1);   // device index of the cut-level ancestor (or self)
      std::vector<double> wsub(nb, 0.0);
      for (int i = 0; i < nb; ++i) {
        const Blk &B = h->blks[i];
        if (B.level < cut) continue;
        int r = -1;
        if (B.level == cut) r = i;
        else if (B.nanc > 0) {
          // through the DIRECT parent (the last ancestor), whose own root is known already: device order sorts blocks by level.
          // Works for make_edges' full ancestor lists and for make_edges_limited's single parents (tree_dep.cpp:133-186) alike
          const int par = h->anc_idx[B.anc_ptr + B.nanc - 1];
          r = h->blks[par].level == cut ? par : root_of[par];
        }
        if (r < 0) return fail_create(h, ST_ERR_TOPOLOGY, "block below the cut level without an ancestor on it");
        root_of[i] = r;
        wsub[r] += (double)B.m * ((double)B.P * B.P + 1.0);
      }
      std::vector<int> roots;
      double tot = 0;
      for (int i = 0; i < nb; ++i) if (h->blks[i].level == cut && h->blks[i].nobs > 0) { roots.push_back(i); tot += wsub[i]; }
      double acc_w = 0;
      std::vector<int> root_owner(nb, 0);
      for (size_t k = 0; k < roots.size(); ++k) {
        int r = (int)std::floor((acc_w + 0.5 * wsub[roots[k]]) / tot * h->world);
        r = std::min(std::max(r, 0), h->world - 1);
        if (k > 0) r = std::max(r, root_owner[roots[k - 1]]);   // keep runs contiguous
        root_owner[roots[k]] = r;
        acc_w += wsub[roots[k]];
      }
      for (int i = 0; i < nb; ++i) if (root_of[i] >= 0) h->blk_owner[i] = root_owner[root_of[i]];
    }
  }
  if (plan_only) {
    if (owner_out) for (int i = 0; i < nb; ++i) owner_out[h->blks[i].model_id] = h->blk_owner[i];
    if (cut_out) *cut_out = h->cut;
    delete h;
    return ST_OK;
  }

  // ---- level lists (u_by_block_groups) and per-level launch geometry
  hipDeviceProp_t prop;
  if (hipSetDevice(h->device) != hipSuccess || hipGetDeviceProperties(&prop, h->device) != hipSuccess)
    return fail_create(h, ST_ERR_HIP, "no usable HIP device (the product path has no CPU fallback)");
  h->sm_count = prop.multiProcessorCount;
  {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, h->device) == hipSuccess && v > 0) h->lds_limit = (size_t)v;
    if (h->lds_limit > 160 * 1024) h->lds_limit = 160 * 1024;
    h->quad_nu = 4;   // units per workgroup of k_factor_quad (2 per workgroup with two workgroups per CU measured slower)
    { const char *e = getenv("SPAMTREE_WIDE"); h->wide_on = (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : 1); }
    { const char *e = getenv("SPAMTREE_LCHAIN"); h->lchain_on = (e && e[0] == '0') ? 0 : 1; }
    { const char *e = getenv("SPAMTREE_LCHAIN_REF"); h->lchain_ref_on = (e && e[0] == '0') ? 0 : 1; }
    { const char *e = getenv("SPAMTREE_LCHAIN_REF_MIN"); h->lchain_ref_min = e ? atoi(e) : 1; }
    { const char *e = getenv("SPAMTREE_GRAM_BIG"); h->gram_big = (e && e[0] == '0') ? 0 : 1; }
  }
  h->levels.resize(n_actual);
  auto geometry = [&](LevelInfo &L, const std::vector<int> &list, bool is_pred) {
    for (int b : list) {
      const Blk &B = h->blks[b];
      L.maxP = std::max(L.maxP, B.P); L.maxM = std::max(L.maxM, B.m); L.maxLd = std::max(L.maxLd, B.ld); L.maxJ = std::max(L.maxJ, B.nanc);
      for (int t = 0; t < B.nanc; ++t) L.maxMa = std::max(L.maxMa, h->blks[h->anc_idx[B.anc_ptr + t]].m);
      // algorithmic bytes / flops, SURVEY.md section 8d
      const double m = B.m, P = B.P, tri = P * (P + 1) / 2, rim = B.isref ? m * (m + 1) / 2 : m;
      double trisum = 0;
      for (int t = 0; t < B.nanc; ++t) { const double ma = h->blks[h->anc_idx[B.anc_ptr + t]].m; trisum += ma * (ma + 1) / 2; }
      if (!is_pred) {
        L.alg_bytes_A += (8.0 * 2 + 8) * (m + P) + (h->q > 1 ? 4 * (m + P) : 0) + 8 * tri + 8 * m * P + 8 * rim + 16;
        L.alg_bytes_B += 8 * m * P + 8 * rim + 8 * P + 40 * m;
        L.alg_bytes_C += 8 * m * P + 8 * rim + 8 * (m + P) + 8;
        L.alg_bytes_msg += 2 * 8 * (P + trisum);
        L.flops_A += 2 * m * P * P + (B.isref ? 2 * m * m * P + m * m * m : 0);
        L.flops_B += (B.isref ? 2.0 / 3 * m * m * m : 0) + 4 * m * P;
        for (int t = 0; t < B.nanc; ++t) { const double ma = h->blks[h->anc_idx[B.anc_ptr + t]].m; L.flops_B += 2 * ma * ma * m; }
        L.flops_C += 2 * m * P + (B.isref ? m * m : 0);
      }
    }
    L.maxMa = std::max(L.maxMa, 1);
    const int SR = 8;
    L.lds_factor = lds_factor_bytes(L.maxP, L.maxM, L.maxMa, SR, false);
    L.big_factor = h->force_generic || L.lds_factor > h->lds_limit;
    if (L.big_factor) L.lds_factor = lds_factor_bytes(L.maxP, L.maxM, L.maxMa, 4, true);
    L.lds_sample = lds_sample_bytes(L.maxP, L.maxM, L.maxLd, false);
    L.big_sample = h->force_generic || L.lds_sample > h->lds_limit || (L.isref && L.maxM > 32 && L.maxM <= 80);   // wide reference blocks: the
    // scratch-arena kernel has the blocked matrix-core solve (the LDS-panel kernel factorises with three barriers per pivot)
    if (L.big_sample) {
      L.lds_sample = lds_sample_bytes(L.maxP, L.maxM, L.maxLd, true);
      // the posterior precision in LDS, factorised and solved by ONE wave without workgroup barriers (wave_chol_solve_lds):
      // every reference level where it fits (config #4: 74 KB, two workgroups per CU).  [The earlier LDS variant -- S and
      // chol(S)^-1, 100 KB, one barrier per pivot -- only paid on levels of at most 2 x CUs blocks.]
      if (L.isref && L.maxM <= 80 && L.lds_sample + lds_sample_sq_bytes(L.maxM) <= h->lds_limit) {
        L.lds_sample += lds_sample_sq_bytes(L.maxM); L.sample_sq = true;
      }
    }
    L.lds_loglik = lds_loglik_bytes(L.maxP, L.maxM);
  };
  for (int g = 0; g < n_actual; ++g) {
    LevelInfo &L = h->levels[g];
    L.first = (int)h->lvl_list.size();
    L.isref = (int)pb->res_is_ref[g];
    // reference order inside a level: block_names order (make_gibbs_groups :238-246); order is irrelevant on device
    for (long long i = 0; i < nb; ++i) {
      const long long u = pb->block_names[i] - 1;
      if (u < 0 || u >= nb) return fail_create(h, ST_ERR_TOPOLOGY, "block_names out of range");
      if (grp_of[u] == g && obs_of[u] > 0) h->lvl_list.push_back(h->blk_model2dev[u]);
    }
    L.count = (int)h->lvl_list.size() - L.first;
    std::sort(h->lvl_list.begin() + L.first, h->lvl_list.end());
    std::vector<int> list(h->lvl_list.begin() + L.first, h->lvl_list.end());
    geometry(L, list, false);
    // column groups for the MFMA path: a reference block alone, or consecutive sibling non-reference blocks
    {
      L.grp_first = (int)h->grps.size();
      bool ok = !h->force_generic && L.maxP <= 256 && L.maxMa <= 32;
      int maxM = 0, maxKb = 0, maxSub = 1;
      size_t i = 0;
      while (ok && i < list.size()) {
        const Blk &B = h->blks[list[i]];
        Grp G;
        G.row0 = B.row0; G.blk0 = list[i]; G.nblk = 1; G.M = B.m; G.P = B.P;
        if (B.m > 32) { ok = false; break; }
        size_t j = i + 1;
        if (!B.isref) {
          const int lastp = B.nanc ? h->anc_idx[B.anc_ptr + B.nanc - 1] : -1;
          while (j < list.size() && G.nblk < 32) {
            const Blk &C = h->blks[list[j]];
            const int lp = C.nanc ? h->anc_idx[C.anc_ptr + C.nanc - 1] : -1;
            if (C.isref || lp != lastp || list[j] != list[j - 1] + 1 || G.M + C.m > 32) break;
            G.M += C.m; G.nblk += 1; ++j;
          }
        }
        maxM = std::max(maxM, G.M);
        for (int t = 0; t < B.nanc; ++t) {
          const int ma = h->blks[h->anc_idx[B.anc_ptr + t]].m;
          maxSub = std::max(maxSub, ma > 16 ? (ma + 1) / 2 : ma);
        }
        maxKb = std::max(maxKb, B.P);
        h->grps.push_back(G);
        i = j;
      }
      L.grp_count = (int)h->grps.size() - L.grp_first;
      if (ok) {
        L.Pm4 = (L.maxP + 3) & ~3;
        L.ldKV = std::max(2, (maxM + 1) & ~1);
        int ldS = std::max(2, maxKb + 4);                       // 4 zero-filled pad columns per staged row
        while ((ldS & 1) || ((ldS >> 1) & 1) == 0) ++ldS;   // 2 * odd: conflict-free A-operand reads
        L.ldS = ldS; L.SRm = maxSub;
        size_t st = (size_t)L.SRm * L.ldS + 16;
        st = std::max(st, (size_t)2 * L.Pm4 + L.Pm4 / 2 + 2);   // prologue alias: ancestor x, y, outcome ids
        st = std::max(st, (size_t)2 * 32 * CH_LD + 216 + 36);     // epilogue alias: R, Ri (stride CH_LD), elimination scratch
        st = ((st + 1) & ~(size_t)1) + (size_t)L.ldS + 16;       // + the zero row at the end
        L.stage_dbl = (int)((st + 1) & ~(size_t)1);
        L.lds_fast = ((size_t)L.Pm4 * L.ldKV + 16 + L.stage_dbl + FM_VPART + 5 * 32) * 8 + 64 * 4 + 64;
        ok = L.lds_fast <= h->lds_limit;
      }
      if (ok) {
        L.Mr4 = std::max(4, (maxM + 3) & ~3);
        L.Mrows = std::max(1, maxM);
        L.ldN = L.maxLd | 1;              // odd stride >= the longest panel row
        int maxJ = 0;
        for (int b : list) maxJ = std::max(maxJ, h->blks[b].nanc);
        L.av_dbl = std::max(32 * maxJ, 224);
        L.maxJ = maxJ;
        const size_t dbl = (size_t)maxM * L.ldN + 32 + (size_t)L.maxP + 32 + 6 * 32 + (size_t)L.av_dbl + 16 + (L.isref ? (size_t)maxM * CH_LD : 0) + 16;
        L.lds_sfast = dbl * 8 + 64 * 4 + 64;
        L.lds_slean = ((size_t)L.maxP + 32 + (size_t)L.av_dbl + 224 + 16 + 7 * 32 + (L.isref ? 2 * 32 * CH_LD : 0) + 16) * 8;
        ok = L.lds_sfast <= h->lds_limit;
      }
      L.fast = ok;
      if (!ok) { h->grps.resize(L.grp_first); L.grp_count = 0; }
      if (!ok && !h->force_generic && L.maxM <= 80 && L.maxP <= BM_MAXP) {   // (a root level, P = 0, included: its 75 x 75 factorisation is the blocked one of the epilogue)
        int ldS = L.maxP + 24;
        while ((ldS & 1) || ((ldS >> 1) & 1) == 0) ++ldS;
        L.bm_ldS = ldS;
        const size_t work = std::max((size_t)17 * ldS + 16 * 80 + BM_KS * 5 * 256, (size_t)2 * L.maxM * L.maxM + 64);   // stage + zero row + V tile + partial V tiles | R, Ri of the epilogue
        L.lds_bigmfma = ((size_t)3 * (L.maxP + L.maxM) + 3 * (size_t)L.maxM + work) * 8 + (size_t)((L.maxP + L.maxM + 1) & ~1) * 4 + 64;
        L.bigmfma = L.lds_bigmfma <= h->lds_limit;
        // non-reference blocks, <= 64 columns, every block behind at least one ancestor: k_factor_lchain (K in registers, the
        // chain factor streamed through LDS twice); a property of the level, the same on every rank
        if (L.bigmfma && h->lchain_on && !h->limited && !L.isref && L.maxM <= 64 && L.maxP <= 544) {
          bool all_anc = true;
          for (int b : list) all_anc = all_anc && h->blks[b].nanc >= 1 && !h->blks[b].isref;
          if (all_anc) L.lchain = L.maxP <= 384 ? 96 : 136;
        }
        // REFERENCE levels behind a chain (round 3): k_factor_lchain for the chain pass (it runs it at more than twice
        // k_factor_bigmfma's rate, and a block's columns are two slabs on two CUs: the single-block top levels gain too), then
        // k_factor_ref_finish per block.  L.count, not the rank's share: a property of the level
        if (L.bigmfma && h->lchain_on && h->lchain_ref_on && !h->limited && L.isref && L.maxM <= 80 && L.maxP <= 544 && L.count >= h->lchain_ref_min) {
          bool all_anc = true;
          for (int b : list) all_anc = all_anc && h->blks[b].nanc >= 1 && h->blks[b].isref;
          if (all_anc) { L.lchain = L.maxP <= 384 ? 96 : 136; L.lchain_ref = true; }
        }
      }
    }
    if (L.lds_factor > h->lds_limit || L.lds_sample > h->lds_limit || L.lds_loglik > h->lds_limit)
      return fail_create(h, ST_ERR_UNSUPPORTED, "block too large for the LDS-resident vectors");
  }
  // direct children that hold a message record: every observed block of a generic level, the first block of each
  // column group of a fast level (the group's record is the sum over its sibling blocks)
  {
    std::vector<char> holder(nb, 0);
    for (int g = 0; g < n_actual; ++g) {
      const LevelInfo &L = h->levels[g];
      if (L.fast) for (int k = 0; k < L.grp_count; ++k) holder[h->grps[L.grp_first + k].blk0] = 1;
      else for (int k = 0; k < L.count; ++k) holder[h->lvl_list[L.first + k]] = 1;
    }
    for (int i = 0; i < nb; ++i) {
      Blk &B = h->blks[i];
      B.dch_ptr = (int)h->dch_idx.size();
      B.ndch = 0;
      if (!dch[i].empty() && !B.isref) return fail_create(h, ST_ERR_TOPOLOGY, "a non-reference block has observed children");
      for (int c : dch[i]) if (holder[c]) { h->dch_idx.push_back(c); B.ndch++; }
      if (B.ndch > 64 && B.nobs > 0 && h->levels[B.level].fast)
        return fail_create(h, ST_ERR_UNSUPPORTED, "more than 64 direct child groups under one block");
    }
  }
  // this rank's runs per level, exchange masks, cut-level record region
  std::vector<unsigned char> rowmask(n, 0), blkmask(nb, 0);
  for (int g = 0; g < n_actual; ++g) {
    LevelInfo &L = h->levels[g];
    L.own_lo = 0; L.own_n = L.count; L.gown_lo = 0; L.gown_n = L.grp_count;
    if (g >= h->cut) {
      int lo = L.count, hi = 0;
      for (int k = 0; k < L.count; ++k)
        if (h->blk_owner[h->lvl_list[L.first + k]] == h->rank) { lo = std::min(lo, k); hi = std::max(hi, k + 1); }
      L.own_lo = lo < hi ? lo : 0; L.own_n = lo < hi ? hi - lo : 0;
      for (int k = L.own_lo; k < L.own_lo + L.own_n; ++k)
        if (h->blk_owner[h->lvl_list[L.first + k]] != h->rank) return fail_create(h, ST_ERR_TOPOLOGY, "a rank's blocks are not contiguous in a level");
      if (L.fast) {
        int glo = L.grp_count, ghi = 0;
        for (int k = 0; k < L.grp_count; ++k) {
          const Grp &Gr = h->grps[L.grp_first + k];
          const bool mine = h->blk_owner[Gr.blk0] == h->rank;
          for (int b2 = 0; b2 < Gr.nblk; ++b2)
            if ((h->blk_owner[Gr.blk0 + b2] == h->rank) != mine) return fail_create(h, ST_ERR_TOPOLOGY, "a column group straddles two ranks");
          if (mine) { glo = std::min(glo, k); ghi = std::max(ghi, k + 1); }
        }
        L.gown_lo = glo < ghi ? glo : 0; L.gown_n = glo < ghi ? ghi - glo : 0;
      }
    }
    // sibling groups for k_factor_wide (levels on the wide-block path): consecutive blocks of this rank's run with the same
    // last parent (= the same chain; device order keeps siblings and their rows contiguous), at most WG_MAXB blocks and
    // WG_MAXN columns per group
    L.wide_first = (int)h->wgrps.size(); L.wide_count = 0; L.wide_maxN = 0;
    // measured at config #4 (577^2 x 3 outcomes): the leaf level 32.7 -> 28.2 ms, the 75-column reference level 13.9 -> 15.1 ms
    // (two blocks per group: more passes than staging saved), levels with fewer groups than CUs lose parallelism -- so only
    // big non-reference levels take it (SPAMTREE_WIDE=2 forces it on every eligible level: tests)
    if (L.bigmfma && !L.lchain && h->wide_on && !h->limited && (h->wide_on == 2 || (!L.isref && L.count >= 2 * h->sm_count))) {   // L.count, not the rank's share: the two kernels round
      // differently, and a level must take the same one on every rank of every world size (bit-identical sharded runs)
      int k = L.own_lo;
      const int kend = L.own_lo + L.own_n;
      while (k < kend) {
        const int b0 = h->lvl_list[L.first + k];
        const Blk &B0 = h->blks[b0];
        const int lastp = B0.nanc ? h->anc_idx[B0.anc_ptr + B0.nanc - 1] : -1;
        WideGrp Gw; Gw.first = k - L.own_lo; Gw.count = 1;
        int Ncols = B0.m;
        while (k + Gw.count < kend && Gw.count < WG_MAXB) {
          const int b1 = h->lvl_list[L.first + k + Gw.count];
          const Blk &B1 = h->blks[b1];
          const int lp1 = B1.nanc ? h->anc_idx[B1.anc_ptr + B1.nanc - 1] : -2;
          if (lp1 != lastp || B1.nanc != B0.nanc || B1.isref != B0.isref || b1 != b0 + Gw.count || Ncols + B1.m > WG_MAXN ||
              B1.row0 != B0.row0 + Ncols) break;
          Ncols += B1.m; ++Gw.count;
        }
        L.wide_maxN = std::max(L.wide_maxN, Ncols);
        h->wgrps.push_back(Gw);
        ++L.wide_count;
        k += Gw.count;
      }
      int ldS = L.maxP + 24;
      while ((ldS & 1) || ((ldS >> 1) & 1) == 0) ++ldS;
      L.bm_ldS = ldS;
      const size_t work = std::max((size_t)17 * ldS + 16 * 16 * WG_JT, (size_t)2 * L.maxM * L.maxM + 64);
      L.lds_wide = ((size_t)3 * (L.maxP + L.wide_maxN) + 2 * (size_t)L.wide_maxN + work) * 8 + (size_t)((L.maxP + L.wide_maxN + 1) & ~1) * 4 + 64;
      if (L.lds_wide > h->lds_limit) { h->wgrps.resize(L.wide_first); L.wide_count = 0; }
    }
    // slabs for k_factor_lchain: sibling groups (consecutive blocks of this rank's run with the same last parent, contiguous
    // rows AND panels, one row stride) cut into runs of <= 4 column tiles, as equal as possible (9 tiles -> 3 + 3 + 3)
    L.lc_first = (int)h->lcslabs.size(); L.lc_count = 0;
    L.rf_first = (int)h->rfvoff.size();
    if (L.lchain) {
      int k = L.own_lo;
      const int kend = L.own_lo + L.own_n;
      long long vrun = 0;   // reference levels: the groups' V matrices (P x the group's columns, row-major) follow each other
      while (k < kend) {
        const int b0 = h->lvl_list[L.first + k];
        const Blk &B0 = h->blks[b0];
        const int lastp = h->anc_idx[B0.anc_ptr + B0.nanc - 1];
        int cnt = 1, Ncols = B0.m;
        while (k + cnt < kend && cnt < 16) {
          const int b1 = h->lvl_list[L.first + k + cnt];
          const Blk &B1 = h->blks[b1];
          if (b1 != b0 + cnt || B1.nanc != B0.nanc || h->anc_idx[B1.anc_ptr + B1.nanc - 1] != lastp || B1.P != B0.P || B1.ld != B0.ld ||
              B1.row0 != B0.row0 + Ncols || B1.panel_off != B0.panel_off + (long long)Ncols * B0.ld) break;
          Ncols += B1.m; ++cnt;
        }
        const int JT = (Ncols + 15) / 16, nsl = (JT + 3) / 4, tps = (JT + nsl - 1) / nsl;
        for (int s0 = 0; s0 < Ncols; s0 += 16 * tps) {
          LcSlab S;
          S.row0 = B0.row0 + s0; S.pan0 = B0.panel_off + (long long)s0 * B0.ld; S.blk0 = b0;
          S.ncol = std::min(16 * tps, Ncols - s0); S.ld = B0.ld; S.vcol0 = s0; S.vs0 = vrun;
          h->lcslabs.push_back(S);
          ++L.lc_count;
        }
        if (L.lchain_ref) {   // the group's blocks are equally wide (same P and ld): block i of the group owns columns [i m, (i + 1) m)
          for (int i = 0; i < cnt; ++i) { h->rfvoff.push_back(vrun); vrun += rf_vsize(B0.P); }
        }
        k += cnt;
      }
      if (L.lchain_ref) h->vscr_need = std::max(h->vscr_need, (size_t)vrun);
    }
    // quads for k_factor_quad: runs of up to quad_nu column groups of one rank that share their ancestor chain
    // (reference levels) or the chain without its last ancestor (leaf levels: cousins)
    L.quad_first = (int)h->quads.size(); L.quad_count = 0; L.q_nkx = 0; L.qown_lo = 0; L.qown_n = 0;
    if (L.fast && L.maxP > 0 && L.maxP <= 200 && L.maxMa <= 32) {
      bool mixed = false;
      int qlo = INT_MAX, qhi = 0;
      // pass 0 ignores ownership: its quad count decides eligibility, so that every rank of every world size takes
      // the same kernel for a level (results are then bit-identical across world sizes); pass 1 builds this rank's quads
      int nq_any = 0;
      // units per workgroup on THIS rank: a sharded level with few owned groups takes smaller quads, so that its
      // workgroups still cover the CUs (a workgroup of 2 / 1 units lives about 0.72 / 0.5 as long as one of 4; results do
      // not depend on the grouping: every unit's arithmetic is its own)
      int nu_max = h->quad_nu;
      {
        int owned = 0;
        for (int k2 = 0; k2 < L.grp_count; ++k2) {
          const Grp &Gq = h->grps[L.grp_first + k2];
          if (g < h->cut || h->blk_owner[Gq.blk0] == h->rank) ++owned;
        }
        double best = 1e300;
        const int cand[3] = {4, 2, 1};
        const double tl[3] = {1.0, 0.72, 0.5};
        for (int c = 0; c < 3; ++c) {
          if (cand[c] > h->quad_nu) continue;
          const double rounds = std::ceil((double)std::max(owned, 1) / (double)(cand[c] * h->sm_count));
          if (rounds * tl[c] < best - 1e-9) { best = rounds * tl[c]; nu_max = cand[c]; }
        }
        const char *e = getenv("SPAMTREE_QUAD_UNITS");   // tests: force the units per workgroup (1, 2 or 4)
        if (e && atoi(e) >= 1 && atoi(e) <= h->quad_nu) nu_max = atoi(e);
      }
      for (int pass = 0; pass < 2; ++pass) {
        int k = 0;
        while (k < L.grp_count) {
          const Grp &G0 = h->grps[L.grp_first + k];
          const Blk &B0 = h->blks[G0.blk0];
          const int J = B0.nanc, Jc = B0.isref ? J : std::max(J - 1, 0);
          if ((B0.isref != 0) != (L.isref != 0)) mixed = true;
          Quad Qd;
          Qd.g0 = k; Qd.nu = 1; Qd.Jc = Jc; Qd.Pc = 0;
          for (int t = 0; t < Jc; ++t) Qd.Pc += h->blks[h->anc_idx[B0.anc_ptr + t]].m;
          while (Qd.nu < (pass == 0 ? h->quad_nu : nu_max) && k + Qd.nu < L.grp_count) {
            const Grp &G1 = h->grps[L.grp_first + k + Qd.nu];
            const Blk &B1 = h->blks[G1.blk0];
            if (B1.nanc != J || B1.isref != B0.isref) break;
            if (pass == 1 && h->blk_owner[G1.blk0] != h->blk_owner[G0.blk0]) break;
            bool same = true;
            for (int t = 0; t < Jc && same; ++t) same = h->anc_idx[B1.anc_ptr + t] == h->anc_idx[B0.anc_ptr + t];
            if (!same) break;
            ++Qd.nu;
          }
          if (pass == 0) ++nq_any;
          else {
            const bool mine = g < h->cut || h->blk_owner[G0.blk0] == h->rank;
            if (mine) { qlo = std::min(qlo, L.quad_count); qhi = std::max(qhi, L.quad_count + 1); }
            h->quads.push_back(Qd);
            L.quad_count++;
          }
          k += Qd.nu;
        }
      }
      L.qown_lo = qlo < qhi ? qlo : 0; L.qown_n = qlo < qhi ? qhi - qlo : 0;
      const int need = (L.maxP + 3) / 4;
      L.q_nkx = need <= 32 ? 32 : (need <= 38 ? 38 : (need <= 44 ? 44 : 50));
      const int ldS = quad_lds_stride(L.q_nkx);   // the kernel's compile-time row stride (>= maxP + 24)
      L.q_ldS = ldS;
      L.lds_quad = ((size_t)h->quad_nu * 16 * ldS + ldS + (L.isref ? (size_t)h->quad_nu * 512 : (size_t)2 * h->quad_nu * QUAD_LEAF_KH * 64)) * 8;   // arena, zero row, V exchange / covariance scratch
      int min_groups = 2 * h->sm_count;   // smaller levels do not fill the chip with quads: k_factor_mfma's 4x more workgroups win
      { const char *e = getenv("SPAMTREE_QUAD_MIN"); if (e) min_groups = atoi(e); }
      if (L.grp_count < 2 * nq_any || mixed || L.grp_count < min_groups) L.q_nkx = 0;   // mostly singletons: nothing to share
      if (L.isref && L.q_nkx == 50) L.q_nkx = 0;                   // that instantiation spills registers: k_factor_mfma is faster
    }
  }
  for (int i = 0; i < nb; ++i) {
    const Blk &B = h->blks[i];
    const bool mine = (h->blk_owner[i] == h->rank) || (h->blk_owner[i] < 0 && h->rank == 0);
    blkmask[i] = mine ? 1 : 0;
    if (mine) for (int r2 = 0; r2 < B.m; ++r2) rowmask[B.row0 + r2] = 1;
    if (B.nobs > 0 && (h->blk_owner[i] < 0 || h->blk_owner[i] == h->rank)) h->own_obs_list.push_back(i);
  }
  if (h->cut < n_actual) {
    const LevelInfo &L = h->levels[h->cut];
    long long lo = -1, hi = -1;
    for (int k = 0; k < L.count; ++k) {
      const Blk &B = h->blks[h->lvl_list[L.first + k]];
      if (lo < 0) lo = B.acc_off;
      hi = B.acc_off + B.acc_len;
      if (h->blk_owner[h->lvl_list[L.first + k]] != h->rank && B.acc_len > 0) h->top_zero.push_back({B.acc_off, (long long)B.acc_len});
    }
    h->top_off = std::max(0LL, lo); h->top_len = hi > lo ? hi - lo : 0;
  }
  for (int i = 0; i < nb; ++i) {
    if (h->blks[i].nobs > 0) h->all_obs_list.push_back(i);
    else {
      if (h->blks[i].nanc == 0) return fail_create(h, ST_ERR_TOPOLOGY, "prediction block without parents");
      h->pred_list.push_back(i);
    }
  }
  geometry(h->pred_info, h->pred_list, true);
  // ---- phase P on the leaf path of k_factor_quad (prediction blocks are non-reference blocks behind a chain of reference
  // blocks, exactly what a leaf level is: spamtree_model.cpp:1296-1326 is A7 + a draw): column groups of consecutive sibling
  // prediction blocks (<= 32 columns) and quads of up to four groups that share all but the last ancestor.  Every rank predicts
  // every block (as the generic kernel does: w is replicated).  Not eligible (long chains, wide blocks): the generic kernel.
  h->pred_grp_first = (int)h->grps.size(); h->pred_grp_count = 0; h->pred_quad_first = (int)h->quads.size(); h->pred_quad_count = 0; h->pred_nkx = 0;
  {
    const LevelInfo &Lp = h->pred_info;
    const char *e = getenv("SPAMTREE_PREDICT_FAST");
    bool ok = !(e && e[0] == '0') && !h->force_generic && !h->pred_list.empty() && Lp.maxP > 0 && Lp.maxP <= 200 && Lp.maxMa <= 32 && Lp.maxM <= 32;
    const std::vector<int> &list = h->pred_list;
    size_t i = 0;
    while (ok && i < list.size()) {
      const Blk &B = h->blks[list[i]];
      if (B.isref || B.nanc < 1) { ok = false; break; }
      Grp G;
      G.row0 = B.row0; G.blk0 = list[i]; G.nblk = 1; G.M = B.m; G.P = B.P;
      const int lastp = h->anc_idx[B.anc_ptr + B.nanc - 1];
      size_t j = i + 1;
      while (j < list.size() && G.nblk < 32) {
        const Blk &C = h->blks[list[j]];
        const int lp = C.nanc ? h->anc_idx[C.anc_ptr + C.nanc - 1] : -1;
        if (C.isref || lp != lastp || list[j] != list[j - 1] + 1 || G.M + C.m > 32 || C.row0 != G.row0 + G.M || C.P != B.P) break;
        G.M += C.m; G.nblk += 1; ++j;
      }
      h->grps.push_back(G);
      i = j;
    }
    if (!ok) h->grps.resize(h->pred_grp_first);
    h->pred_grp_count = (int)h->grps.size() - h->pred_grp_first;
    if (ok && h->pred_grp_count > 0) {
      int k = 0;
      while (k < h->pred_grp_count) {
        const Grp &G0 = h->grps[h->pred_grp_first + k];
        const Blk &B0 = h->blks[G0.blk0];
        const int J = B0.nanc, Jc = J - 1;
        Quad Qd;
        Qd.g0 = k; Qd.nu = 1; Qd.Jc = Jc; Qd.Pc = 0;
        for (int t = 0; t < Jc; ++t) Qd.Pc += h->blks[h->anc_idx[B0.anc_ptr + t]].m;
        while (Qd.nu < h->quad_nu && k + Qd.nu < h->pred_grp_count) {
          const Blk &B1 = h->blks[h->grps[h->pred_grp_first + k + Qd.nu].blk0];
          if (B1.nanc != J) break;
          bool same = true;
          for (int t = 0; t < Jc && same; ++t) same = h->anc_idx[B1.anc_ptr + t] == h->anc_idx[B0.anc_ptr + t];
          if (!same) break;
          ++Qd.nu;
        }
        h->quads.push_back(Qd);
        k += Qd.nu;
      }
      h->pred_quad_count = (int)h->quads.size() - h->pred_quad_first;
      const int need = (Lp.maxP + 3) / 4;
      h->pred_nkx = need <= 32 ? 32 : (need <= 38 ? 38 : (need <= 44 ? 44 : 50));
      h->pred_lds = ((size_t)h->quad_nu * 16 * quad_lds_stride(h->pred_nkx) + quad_lds_stride(h->pred_nkx) + (size_t)2 * h->quad_nu * QUAD_LEAF_KH * 64) * 8;
    }
  }

  // ---- row data in device order
  std::vector<double> cx(n), cy(n), y(n), X((size_t)n * pb->p);
  std::vector<int> mv(n);
  std::vector<unsigned char> obs(n);
  h->n_obs_q.assign(pb->q, 0);
  for (long long i = 0; i < n; ++i) {
    const long long r = h->dev2model[i];
    cx[i] = pb->coords[r]; cy[i] = pb->coords[n + r];
    const long long v = pb->mv_id[r] - 1;
    if (v < 0 || v >= pb->q) return fail_create(h, ST_ERR_USAGE, "mv_id out of range");
    mv[i] = (int)v;
    const bool ok = std::isfinite(pb->y[r]);
    obs[i] = ok ? 1 : 0;
    y[i] = ok ? pb->y[r] : 0.0;                                  // spamtree_model.cpp:146
    if (ok) { h->n_obs_q[v]++; h->n_obs++; }
    for (int j = 0; j < pb->p; ++j) X[(size_t)j * n + i] = pb->X[(size_t)j * n + r];
  }
  // Q3 partner rows (spamtree_model.cpp:1375): the t-th available row is paired with w[t]
  std::vector<long long> partner(n);
  {
    std::vector<long long> rank_av(n, -1);
    long long t = 0;
    for (long long r = 0; r < n; ++r)
      if (std::isfinite(pb->y[r])) rank_av[r] = t++;
    for (long long i = 0; i < n; ++i) {
      const long long r = h->dev2model[i];
      partner[i] = (h->quirks && rank_av[r] >= 0) ? h->model2dev[rank_av[r]] : i;
    }
  }
  // XtX(j) over observed rows of outcome j (:151-155)
  h->xtx.assign((size_t)pb->p * pb->p * pb->q, 0.0);
  for (long long r = 0; r < n; ++r) {
    if (!std::isfinite(pb->y[r])) continue;
    const int v = (int)(pb->mv_id[r] - 1);
    for (int a = 0; a < pb->p; ++a)
      for (int b2 = 0; b2 < pb->p; ++b2)
        h->xtx[(size_t)v * pb->p * pb->p + (size_t)b2 * pb->p + a] += pb->X[(size_t)a * n + r] * pb->X[(size_t)b2 * n + r];
  }

#define CCHK(call)                                                                                            \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) return fail_create(h, ST_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
  CCHK(hipStreamCreate(&h->stream));
  CCHK(hipHostMalloc((void **)&h->pin, 64 * sizeof(double), hipHostMallocDefault));
  h->pin_up_len = QMAX + std::max(1, h->p * h->q);
  CCHK(hipHostMalloc((void **)&h->pin_up, (size_t)2 * h->pin_up_len * sizeof(double), hipHostMallocDefault));
  for (int i = 0; i < 2; ++i) CCHK(hipEventCreateWithFlags(&h->ev_up[i], hipEventDisableTiming));
  CCHK(hipEventCreateWithFlags(&h->ev_factor, hipEventDisableTiming));
  CCHK(h->d_cx.upload(cx)); CCHK(h->d_cy.upload(cy)); CCHK(h->d_y.upload(y)); CCHK(h->d_X.upload(X));
  CCHK(h->d_mv.upload(mv)); CCHK(h->d_obs.upload(obs)); CCHK(h->d_partner.upload(partner));
  CCHK(h->d_dev2model.upload(h->dev2model));
  {
    // device copy: acc_len = where, inside a child's record, the part FOR this block starts (after the block's own
    // ancestors in the full tree; at 0 when every block has a single parent)
    std::vector<Blk> db = h->blks;
    if (h->limited) for (Blk &B : db) B.acc_len = 0;
    CCHK(h->d_blks.upload(db));
    if (h->limited) { std::vector<int> t = h->twin_list; if (t.empty()) t.push_back(0); CCHK(h->d_twin.upload(t)); }
  }
  { std::vector<int> a = h->anc_idx; if (a.empty()) a.push_back(0); CCHK(h->d_anc.upload(a)); }
  { std::vector<int> a = h->dch_idx; if (a.empty()) a.push_back(0); CCHK(h->d_dch.upload(a)); }
  CCHK(h->d_lvl.upload(h->lvl_list));
  { std::vector<Grp> g = h->grps; if (g.empty()) g.push_back(Grp{0, 0, 0, 0, 0}); CCHK(h->d_grps.upload(g)); }
  {
    // group descriptors: the flattened metadata of every column group (layout: GdHead / gd_unpack)
    int stride = 8;
    for (const Grp &G : h->grps) {
      const Blk &B0 = h->blks[G.blk0];
      stride = std::max(stride, 8 + 4 * B0.nanc + 3 * G.nblk + 2 * std::min(B0.ndch, 64));   // children: record offset + group id
    }
    stride = (stride + 1) & ~1;
    if (stride > GD_MAXW) return fail_create(h, ST_ERR_UNSUPPORTED, "group descriptor too long");
    h->gd_stride = stride;
    h->gdesc.assign(std::max<size_t>(1, h->grps.size()) * (size_t)stride, 0);
    auto pack = [](long long lo, long long hi) { return (lo & 0xffffffffLL) | (hi << 32); };
    std::vector<long long> blk2grp((size_t)nb, -1);   // the group that holds a block (its first block holds the group's record)
    for (size_t g = 0; g < h->grps.size(); ++g)
      for (int b = 0; b < h->grps[g].nblk; ++b) blk2grp[h->grps[g].blk0 + b] = (long long)g;
    for (size_t g = 0; g < h->grps.size(); ++g) {
      const Grp &G = h->grps[g];
      const Blk &B0 = h->blks[G.blk0];
      long long *w = h->gdesc.data() + g * (size_t)stride;
      const int nch = std::min(B0.ndch, 64);
      w[0] = G.row0; w[1] = B0.acc_off; w[2] = pack(G.M, G.P); w[3] = pack(B0.nanc, G.nblk); w[4] = pack(B0.isref, B0.level);
      w[5] = pack(nch, h->limited ? 0 : B0.acc_len); w[6] = pack(G.blk0, 0);
      long long ao = 0, aoff = 0;
      for (int t = 0; t < B0.nanc; ++t) {
        const Blk &Ba = h->blks[h->anc_idx[B0.anc_ptr + t]];
        long long *a = w + 8 + 4 * t;
        a[0] = pack(Ba.m, ao); a[1] = Ba.row0; a[2] = Ba.chain_off; a[3] = aoff;
        ao += Ba.m; aoff += (long long)Ba.m * Ba.m + Ba.m;
      }
      w[7] = aoff;
      for (int b = 0; b < G.nblk; ++b) {
        const Blk &Bb = h->blks[G.blk0 + b];
        long long *q = w + 8 + 4 * B0.nanc + 3 * b;
        q[0] = Bb.panel_off; q[1] = Bb.row0; q[2] = Bb.ld;
      }
      for (int c = 0; c < nch; ++c) w[8 + 4 * B0.nanc + 3 * G.nblk + c] = h->blks[h->dch_idx[B0.dch_ptr + c]].acc_off;
      for (int c = 0; c < nch; ++c) w[8 + 4 * B0.nanc + 3 * G.nblk + nch + c] = blk2grp[h->dch_idx[B0.dch_ptr + c]];   // k_gram_direct
    }
    // Gram parts of the last reference level straight from the leaf groups' panels (k_gram_direct): every block of that level
    // has at most GRAM_DIRECT_MAXCH children, all of them column groups of the (non-reference) last level
    h->gram_direct_level = -1;
    {
      const int gl = n_actual - 1, gp = n_actual - 2;
      const char *e = getenv("SPAMTREE_GRAM_DIRECT");
      if (!(e && e[0] == '0') && gp >= 0 && !h->limited && h->levels[gl].fast && !h->levels[gl].isref && h->levels[gp].fast && h->levels[gp].isref &&
          h->levels[gl].maxM <= 32 && h->levels[gl].maxP <= 255) {
        bool ok = true;
        const LevelInfo &Lp = h->levels[gp], &Ll = h->levels[gl];
        for (int k = 0; k < Lp.grp_count && ok; ++k) {
          const Grp &G = h->grps[Lp.grp_first + k];
          const Blk &B0 = h->blks[G.blk0];
          if (G.nblk != 1 || B0.ndch > GRAM_DIRECT_MAXCH) ok = false;
          for (int c = 0; c < B0.ndch && ok; ++c) {
            const long long cg = blk2grp[h->dch_idx[B0.dch_ptr + c]];
            if (cg < Ll.grp_first || cg >= Ll.grp_first + Ll.grp_count || h->grps[cg].M > 32) ok = false;
          }
        }
        // ... and every leaf group's record is read by a block of that level only (its direct parent)
        if (ok) h->gram_direct_level = gp;
      }
    }
    CCHK(h->d_gdesc.upload(h->gdesc));
  }
  { std::vector<Quad> g = h->quads; if (g.empty()) g.push_back(Quad{0, 0, 0, 0}); CCHK(h->d_quads.upload(g)); }
  { std::vector<WideGrp> g = h->wgrps; if (g.empty()) g.push_back(WideGrp{0, 0}); CCHK(h->d_wgrps.upload(g)); }
  if (!h->lcslabs.empty()) { CCHK(h->d_lcslabs.upload(h->lcslabs)); CCHK(h->d_lcrow.alloc((size_t)2 * h->n_all)); }
  if (!h->rfvoff.empty()) { CCHK(h->d_rfvoff.upload(h->rfvoff)); CCHK(h->d_vscr.alloc(h->vscr_need + (size_t)2 * RF_BUFD)); }   // (+ what a chunk's DMA reads past the last block)
  {
    std::vector<long long> s0off((size_t)(nb > 0 ? nb : 1), -1);
    size_t tot = 0;
    for (int g = 0; g < n_actual; ++g) {
      const LevelInfo &L = h->levels[g];
      if (!L.isref || !(L.big_sample || L.fast)) continue;   // generic wide-block levels (round 2) and the column-group levels (round 3)
      for (int k = 0; k < L.count; ++k) {
        const int b = h->lvl_list[L.first + k];
        s0off[b] = (long long)tot;
        tot += (size_t)h->blks[b].m * h->blks[b].m;
      }
    }
    CCHK(h->d_s0off.upload(s0off));
    CCHK(h->d_s0.alloc(std::max(tot, (size_t)1)));
  }
  { std::vector<int> a = h->pred_list; if (a.empty()) a.push_back(0); CCHK(h->d_pred.upload(a)); }
  CCHK(h->d_allobs.upload(h->all_obs_list));
  { std::vector<int> a = h->own_obs_list; if (a.empty()) a.push_back(0); CCHK(h->d_ownobs.upload(a)); }
  for (int g = 0; g < n_actual; ++g) {
    const LevelInfo &L = h->levels[g];
    if (L.fast) for (int k = 0; k < L.gown_n; ++k) h->own_grp_list.push_back(L.grp_first + L.gown_lo + k);
  }
  for (int b : h->own_obs_list) if (!h->levels[h->blks[b].level].fast) h->own_obs_slow.push_back(b);
  { std::vector<int> a = h->own_grp_list; if (a.empty()) a.push_back(0); CCHK(h->d_owngrp.upload(a)); }
  { std::vector<int> a = h->own_obs_slow; if (a.empty()) a.push_back(0); CCHK(h->d_ownslow.upload(a)); }
  CCHK(h->d_rowmask.upload(rowmask)); CCHK(h->d_blkmask.upload(blkmask));
  CCHK(h->d_comm.alloc((size_t)2 * nb + 64));
  {
    // all-gather of w: every rank's owned rows (blocks below the cut, prediction blocks included), in device order; the
    // replicated top is sampled identically everywhere and does not travel
    std::vector<std::vector<int>> rows_of(h->world);
    for (int i = 0; i < nb; ++i) {
      const int o = h->blk_owner[i];
      if (o < 0) continue;
      const Blk &B = h->blks[i];
      for (int r2 = 0; r2 < B.m; ++r2) rows_of[o].push_back((int)(B.row0 + r2));
    }
    size_t mx = 0;
    for (auto &v : rows_of) mx = std::max(mx, v.size());
    h->gather_cnt = (int)mx + 1;   // last slot: the rank's failure word
    std::vector<int> gi((size_t)h->world * h->gather_cnt, -1);
    for (int r = 0; r < h->world; ++r)
      for (size_t i2 = 0; i2 < rows_of[r].size(); ++i2) gi[(size_t)r * h->gather_cnt + i2] = rows_of[r][i2];
    CCHK(h->d_gidx.upload(gi));
    CCHK(h->d_gather.alloc(gi.size()));
    CCHK(h->d_gerr.alloc(64));
  }
  CCHK(h->d_w.alloc(n)); CCHK(h->d_xb.alloc(n)); CCHK(h->d_z.alloc(n)); CCHK(h->d_tmp_n.alloc(n + 64));
  CCHK(hipMemset(h->d_w.p, 0, n * sizeof(double)));
  CCHK(hipMemset(h->d_xb.p, 0, n * sizeof(double)));
  CCHK(hipMemset(h->d_z.p, 0, n * sizeof(double)));
  CCHK(h->d_B.alloc((size_t)pb->p * pb->q));
  CCHK(h->d_tsq.alloc(QMAX));
  for (int s = 0; s < 2; ++s) {
    // + 128: k_factor_lchain stages chain rows in whole 128-double LDS-DMA pieces and may read up to 127 doubles past a row's
    // end (the other kernels: one double, an odd row's 16-byte piece); the padding is zeroed with the arena
    CCHK(h->d_panels[s].alloc(h->panel_total + 128));
    CCHK(hipMemset(h->d_panels[s].p, 0, (h->panel_total + 128) * sizeof(double)));
    CCHK(h->d_logdet[s].alloc(nb)); CCHK(h->d_loglik[s].alloc(nb));
    CCHK(hipMemset(h->d_logdet[s].p, 0, nb * sizeof(double)));
    CCHK(hipMemset(h->d_loglik[s].p, 0, nb * sizeof(double)));
  }
  CCHK(h->d_acc.alloc(std::max<size_t>(h->acc_total, 1)));
  CCHK(hipMemset(h->d_acc.p, 0, std::max<size_t>(h->acc_total, 1) * sizeof(double)));
  CCHK(h->d_scalars.alloc(8 + 2 * SUM2_WG));
  CCHK(h->d_err.alloc(2));
  CCHK(h->d_partial.alloc((size_t)STATS_WG * (pb->p * pb->q + pb->q)));
  CCHK(h->d_stats.alloc((size_t)pb->p * pb->q + pb->q));
  // scratch for the generic kernels: a bounded number of resident workgroups, each with its own slice
  {
    size_t need = 0;
    auto upd = [&](const LevelInfo &L) {
      if (L.big_factor || L.bigmfma) need = std::max(need, scratch_factor_doubles(L.maxP, L.maxM, L.maxMa));
      if (L.wide_count > 0) need = std::max(need, (size_t)2 * L.maxP * L.wide_maxN + (size_t)L.maxMa * L.wide_maxN);
      if (L.big_sample) need = std::max(need, (size_t)L.maxM * L.maxM);
    };
    for (auto &L : h->levels) upd(L);
    if (!h->pred_list.empty()) upd(h->pred_info);
    if (need > 0) {
      h->scratch_wgs = h->sm_count * 4;
      h->scratch_stride = (long long)((need + 15) & ~(size_t)15);
      CCHK(h->d_scratch.alloc((size_t)h->scratch_wgs * h->scratch_stride));
    }
  }
  // opt in to > 64 KiB dynamic LDS
  const int lim = (int)h->lds_limit;
  (void)hipFuncSetAttribute((const void *)k_factor<false, MODE_FACTOR>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor<true, MODE_FACTOR>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor<false, MODE_PREDICT>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor<true, MODE_PREDICT>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_sample<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_sample<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_sample<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_loglik, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor_wide<WG_JT>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  {
    // k_factor_lchain: static + dynamic LDS must fit one CU's 160 KB, else the level stays on the older kernels
    hipFuncAttributes fa;
    size_t st96 = 16 * 1024, st136 = 16 * 1024;
    bool ok96 = false, ok136 = false;   // a build whose unrolling failed keeps K in scratch memory (localSizeBytes > 0): never use that
    if (hipFuncGetAttributes(&fa, (const void *)k_factor_lchain<96>) == hipSuccess) { st96 = fa.sharedSizeBytes; ok96 = fa.localSizeBytes == 0; }
    if (hipFuncGetAttributes(&fa, (const void *)k_factor_lchain<136>) == hipSuccess) { st136 = fa.sharedSizeBytes; ok136 = fa.localSizeBytes == 0; }
    for (auto &L : h->levels) {
      if (!L.lchain) continue;
      const size_t need = lc_dyn_doubles(L.lchain) * 8 + (L.lchain == 96 ? st96 : st136);
      if (need > 160 * 1024 || !(L.lchain == 96 ? ok96 : ok136)) { L.lchain = 0; L.lchain_ref = false; }
      if (L.lchain_ref) {
        L.lds_ref_finish = rf_lds_bytes(L.maxM);
        if (L.lds_ref_finish > h->lds_limit) { L.lchain = 0; L.lchain_ref = false; }
      }
    }
    (void)hipFuncSetAttribute((const void *)k_factor_ref_finish, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void *)k_factor_lchain<96>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lc_dyn_doubles(96) * 8));
    (void)hipFuncSetAttribute((const void *)k_factor_lchain<136>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lc_dyn_doubles(136) * 8));
  }
  (void)hipFuncSetAttribute((const void *)k_factor_bigmfma<5, 3, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor_bigmfma<3, 5, 34>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor_bigmfma<4, 5, 34>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_sample_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_sample_wave, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_marginal_invchol, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_marginal_invchol_wave, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  {
    // phase A kernel for the column-group levels: 3 (default) = k_factor_quad where a level is eligible (big enough,
    // chains <= 200 rows, LDS fits) and k_factor_mfma elsewhere; 1 = k_factor_mfma everywhere
    const char *e = getenv("SPAMTREE_FACTOR_KERNEL");
    h->factor_gen = (e && e[0] == '1') ? 1 : 3;
    { const char *e2 = getenv("SPAMTREE_SAMPLE_LEAN"); h->sample_lean = (e2 && e2[0] == '0') ? 0 : 1; }
    { const char *e2 = getenv("SPAMTREE_LEAF_SEG"); h->leaf_seg = (e2 && e2[0] == '0') ? 0 : 1; }
    { const char *e2 = getenv("SPAMTREE_LEAF_WIDE"); h->leaf_wide = (e2 && e2[0] == '0') ? 0 : 1; }
    { const char *e2 = getenv("SPAMTREE_SPLIT_GRAM"); h->split_gram = (e2 && e2[0] == '0') ? 0 : ((e2 && e2[0] == '2') ? 2 : 1); }
    { const char *e2 = getenv("SPAMTREE_SAMPLE_WAVE"); h->sample_wave = (e2 && e2[0] == '0') ? 0 : ((e2 && e2[0] == '2') ? 2 : 1); }   // 2: every eligible level (tests)
  }
  {
    // k_factor_quad: static + dynamic LDS must fit; levels that do not fit (or are too small to fill the chip) keep k_factor_mfma
    const void *fq = (const void *)k_factor_quad<4, 50, 13, false, true>;
    hipFuncAttributes fa;
    size_t stat = 24 * 1024;
    if (hipFuncGetAttributes(&fa, fq) == hipSuccess) stat = fa.sharedSizeBytes;
    {
      const void *fr = (const void *)k_factor_quad<4, 50, 13, true, true>;
      if (hipFuncGetAttributes(&fa, fr) == hipSuccess) stat = std::max(stat, (size_t)fa.sharedSizeBytes);
      const void *ft = (const void *)k_factor_quad<4, 50, 13, true, false>;
      if (hipFuncGetAttributes(&fa, ft) == hipSuccess) stat = std::max(stat, (size_t)fa.sharedSizeBytes);
    }
    for (auto &L : h->levels) {
      if (L.q_nkx == 0) continue;
      if (L.lds_quad + stat > 160 * 1024) L.q_nkx = 0;
    }
    if (h->pred_nkx && h->pred_lds + stat > 160 * 1024) h->pred_nkx = 0;
#define QATTR(NU_, NKX_, NKT_)                                                                                                       \
  (void)hipFuncSetAttribute((const void *)k_factor_quad<NU_, NKX_, NKT_, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)stat); \
  (void)hipFuncSetAttribute((const void *)k_factor_quad<NU_, NKX_, NKT_, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)stat); \
  (void)hipFuncSetAttribute((const void *)k_factor_quad<NU_, NKX_, NKT_, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)stat)
    QATTR(4, 50, 13); QATTR(4, 44, 11); QATTR(4, 38, 10); QATTR(4, 32, 8);
#undef QATTR
  }
  {
    // top levels that st_factor_begin may run ahead: the leading levels on k_factor_mfma (no global scratch), when every
    // level of the tree is on the column-group path (the generic kernels share one scratch arena between phases)
    h->g_top = 0;
    bool all_fast = !h->limited && !h->force_generic;
    for (int g = 0; g < n_actual; ++g) all_fast = all_fast && h->levels[g].fast;
    if (all_fast) {
      while (h->g_top < n_actual && !(h->factor_gen == 3 && h->levels[h->g_top].q_nkx > 0)) ++h->g_top;
      if (h->g_top >= n_actual) h->g_top = 0;   // nothing would be left for the main stream to hide it under
    }
    // the top levels are a fixed cost (0.19 ms at n = 1e6: a quarter of a rank's phase A on 8 GPUs, 40 % of phase A at
    // n = 1e5); at n = 1e6 on one GPU the sweep fills the chip and the gain is 1.5 % (SPAMTREE_ASYNC_TOP=0 turns it off)
    const char *e = getenv("SPAMTREE_ASYNC_TOP");
    h->async_top = h->g_top > 0 && !(e && e[0] == '0');
    std::vector<int> tl;
    for (int b : h->own_obs_list) if (h->blks[b].level < h->g_top) tl.push_back(b);
    h->n_toplist = (int)tl.size();
    if (tl.empty()) tl.push_back(0);
    CCHK(h->d_toplist.upload(tl));
    CCHK(h->d_err2.alloc(2));
    if (h->async_top) {
      CCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
      CCHK(hipEventCreateWithFlags(&h->ev_top, hipEventDisableTiming));
      CCHK(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
      CCHK(hipEventCreateWithFlags(&h->ev_stats, hipEventDisableTiming));
    }
  }
  (void)hipGetLastError();
#undef CCHK
  h->s0_valid.assign((size_t)std::max(n_actual, 1), 0);
  h->prof_level_ms.assign(2 * n_actual, 0.0);
  h->prof_level_n.assign(2 * n_actual, 0);
  *out = h;
  return ST_OK;
}

// ---- simple state accessors ---------------------------------------------------------------------------------
static int upload_rows(st_handle h, const double *src, double *dst) {
  std::vector<double> tmp(h->n_all);
  for (long long i = 0; i < h->n_all; ++i) tmp[i] = src[h->dev2model[i]];
  HCHK(h, hipMemcpyAsync(dst, tmp.data(), h->n_all * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  return ST_OK;
}
static int download_rows(st_handle h, const double *src, double *dst) {
  std::vector<double> tmp(h->n_all);
  HCHK(h, hipMemcpyAsync(tmp.data(), src, h->n_all * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  for (long long i = 0; i < h->n_all; ++i) dst[h->dev2model[i]] = tmp[i];
  return ST_OK;
}

extern "C" int st_set_w(st_handle h, const double *w) {
  if (!h || !w) return ST_ERR_USAGE;
  invalidate_stats(h);
  HCHK(h, hipSetDevice(h->device));
  return upload_rows(h, w, h->d_w.p);
}
extern "C" int st_get_w(st_handle h, double *w) {
  if (!h || !w) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  return download_rows(h, h->d_w.p, w);
}
extern "C" int st_get_xb(st_handle h, double *xb) {
  if (!h || !xb) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  return download_rows(h, h->d_xb.p, xb);
}
extern "C" int st_set_beta(st_handle h, const double *Bcoeff) {
  if (!h || !Bcoeff) return ST_ERR_USAGE;
  invalidate_stats(h);
  HCHK(h, hipSetDevice(h->device));
  // through the handle's pinned staging (the caller's buffer is free on return; a slot is rewritten only after its last copy
  // has run): NO host synchronisation -- the C++ driver calls the setters while the proposal's factorisation is still running
  // (st_factor_enqueue / st_factor_finish), so that the new XB is in place the moment phase A ends
  h->pin_up_slot ^= 1;
  HCHK(h, hipEventSynchronize(h->ev_up[h->pin_up_slot]));
  double *stg = h->pin_up + (size_t)h->pin_up_slot * h->pin_up_len + QMAX;
  std::memcpy(stg, Bcoeff, (size_t)h->p * h->q * sizeof(double));
  HCHK(h, hipMemcpyAsync(h->d_B.p, stg, (size_t)h->p * h->q * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HCHK(h, hipEventRecord(h->ev_up[h->pin_up_slot], h->stream));
  {
    ProfScope ps(h, 4);
    const int grid = (int)((h->n_all + NT - 1) / NT);
    hipLaunchKernelGGL(k_xb, dim3(grid), dim3(NT), 0, h->stream, h->d_X.p, h->d_mv.p, h->d_B.p, h->n_all, h->p, h->d_xb.p);
  }
  HCHK(h, hipGetLastError());
  return ST_OK;
}
extern "C" int st_set_tausq_inv(st_handle h, const double *t) {
  if (!h || !t) return ST_ERR_USAGE;
  for (int j = 0; j < h->q; ++j) h->tausq_inv[j] = t[j];
  HCHK(h, hipSetDevice(h->device));
  h->pin_up_slot ^= 1;                                      // (pinned staging, no host synchronisation: st_set_beta)
  HCHK(h, hipEventSynchronize(h->ev_up[h->pin_up_slot]));
  double *stg = h->pin_up + (size_t)h->pin_up_slot * h->pin_up_len;
  std::memcpy(stg, h->tausq_inv, QMAX * sizeof(double));
  HCHK(h, hipMemcpyAsync(h->d_tsq.p, stg, QMAX * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HCHK(h, hipEventRecord(h->ev_up[h->pin_up_slot], h->stream));
  return ST_OK;
}
// readers of a slot's arena while st_factor_begin's launches may still be writing it on the second stream: order the
// main stream behind them (top_pending stays set: the next st_factor still picks the result up)
static int settle_top(st_handle h) {
  if (h->top_pending) { HCHK(h, hipSetDevice(h->device)); HCHK(h, hipStreamWaitEvent(h->stream, h->ev_top, 0)); }
  return ST_OK;
}
extern "C" int st_swap(st_handle h) {
  if (!h) return ST_ERR_USAGE;
  if (h->top_pending) {   // the proposal's top levels are being written into the arena that would become the accepted slot
    h->err = "st_swap between st_factor_begin and the st_factor / st_factor_local that picks its result up";
    return ST_ERR_USAGE;
  }
  std::swap(h->slot_map[0], h->slot_map[1]);
  std::swap(h->theta[0], h->theta[1]);
  h->gram_valid = false;
  std::fill(h->s0_valid.begin(), h->s0_valid.end(), 0);
  return ST_OK;
}
extern "C" int st_synchronize(st_handle h) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  HCHK(h, hipStreamSynchronize(h->stream));
  return ST_OK;
}
extern "C" void *st_stream(st_handle h) { return h ? (void *)h->stream : nullptr; }
// run every kernel on the caller's stream (e.g. the stream RCCL collectives are enqueued on): no host synchronisation
// is then needed between the library's kernels and the exchanges
extern "C" int st_set_stream(st_handle h, void *stream) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  HCHK(h, hipStreamSynchronize(h->stream));
  prof_harvest(h);
  if (h->stream && !h->ext_stream) (void)hipStreamDestroy(h->stream);
  h->stream = (hipStream_t)stream;
  h->ext_stream = true;
  return ST_OK;
}
extern "C" int st_shard_info(st_handle h, int32_t *rank, int32_t *world, int32_t *cut_level, int64_t *owned_blocks, int64_t *owned_rows) {
  if (!h) return ST_ERR_USAGE;
  if (rank) *rank = h->rank;
  if (world) *world = h->world;
  if (cut_level) *cut_level = h->cut;
  long long ob = 0, orow = 0;
  for (size_t i = 0; i < h->blks.size(); ++i) if (h->blk_owner[i] == h->rank) { ++ob; orow += h->blks[i].m; }
  if (owned_blocks) *owned_blocks = ob;
  if (owned_rows) *owned_rows = orow;
  return ST_OK;
}

// CovarianceParams::transform (covariance_functions.cpp:34-75) + vec_to_symmat (:77-92)
static int make_covpar(st_handle h, const double *theta, int ntheta, CovPar *cp) {
  const int q = h->q, ncb = q > 2 ? 3 : 1, npars = 3 * q + ncb, k = q * (q - 1) / 2;
  if (ntheta != npars + k) { h->err = "theta has the wrong length"; return ST_ERR_USAGE; }
  std::memset(cp, 0, sizeof(*cp));
  cp->q = q; cp->ncb = ncb;
  for (int j = 0; j < q; ++j) { cp->ai1[j] = theta[j]; cp->ai2[j] = theta[q + j]; cp->phi[j] = theta[2 * q + j]; }
  for (int j = 0; j < ncb; ++j) cp->tmv[j] = theta[3 * q + j];
  int ix = 0;
  for (int j = 0; j < q; ++j)
    for (int i = j + 1; i < q; ++i) { cp->D[i * q + j] = theta[npars + ix]; cp->D[j * q + i] = theta[npars + ix]; ++ix; }
  finish_covpar(cp);
  return ST_OK;
}

template <bool BIG, int MODE>
static void launch_factor(st_handle h, const LevelInfo &L, FactorArgs &A, const CovPar &cp, hipStream_t st = nullptr) {
  int grid = A.nlist;
  if (BIG) {
    grid = std::min(grid, h->scratch_wgs);
    A.scratch = h->d_scratch.p; A.scratch_stride = h->scratch_stride;
  }
  hipLaunchKernelGGL((k_factor<BIG, MODE>), dim3(grid), dim3(NT), L.lds_factor, st ? st : h->stream, A, cp);
}

static int reduce_loglik(st_handle h, int phys, double *loglik) {
  {
    ProfScope ps(h, 3);
    launch_sum2(h->stream, h->d_logdet[phys].p, h->d_loglik[phys].p, (int)h->n_blocks, h->d_scalars.p + 8, h->d_scalars.p);
  }
  HCHK(h, hipGetLastError());
  double s[2];
  HCHK(h, hipMemcpyAsync(s, h->d_scalars.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  *loglik = s[0] + s[1];   // loglik_w = logdetCi + sum(loglik_w_comps)  (:987-988, :815-816)
  return ST_OK;
}

static int read_err(st_handle h, int *code) {
  int e[2];
  HCHK(h, hipMemcpyAsync(e, h->d_err.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  *code = (e[0] == INT_MAX) ? 0 : (e[0] & 15);
  return ST_OK;
}
static int reset_err(st_handle h) {
  const int init[2] = {INT_MAX, 0};
  HCHK(h, hipMemcpyAsync(h->d_err.p, init, 2 * sizeof(int), hipMemcpyHostToDevice, h->stream));
  return ST_OK;
}

// sum-with-zeros exchange buffers: every entry is contributed by exactly one rank (replicated blocks by rank 0), so
// an all-reduce(sum) reproduces the single-GPU arrays bit for bit, for any number of ranks and any reduction order
// all-gather form of the exchange of w: a rank's slice of the gather buffer = its owned rows + its failure word

// levels [g_lo, g_hi); `st` / `errflag`: the launch stream and failure word (st_factor_begin: the second stream, d_err2)
static int factor_launch(st_handle h, int phys, const CovPar &cp, int g_lo = 0, int g_hi = INT_MAX, hipStream_t st = nullptr, int *errflag = nullptr) {
  g_hi = std::min(g_hi, h->n_actual_groups);
  if (!st) st = h->stream;
  if (!errflag) errflag = h->d_err.p;
  int n_launch = 0;
  for (int g = g_lo; g < g_hi; ++g) n_launch += ((h->levels[g].fast ? h->levels[g].gown_n : h->levels[g].own_n) != 0);
  ProfScope phase(h, 0, -2, n_launch, st);   // profile mode 2: the phase's launches between ONE pair of events (mean launch = total / launches)
  if (g_lo == 0 && h->limited && !h->twin_list.empty()) {
    MarginalArgs M;
    M.blks = h->d_blks.p; M.list = h->d_twin.p; M.nlist = (int)h->twin_list.size(); M.cx = h->d_cx.p; M.cy = h->d_cy.p; M.mv = h->d_mv.p;
    M.panels = h->d_panels[phys].p; M.errflag = errflag; M.maxM = h->twin_maxM;
    const size_t lds = (size_t)2 * h->twin_maxM * h->twin_maxM * sizeof(double);
    if (h->twin_maxM <= 27 && !h->force_generic)   // one block per wave, blocked DPP / MFMA elimination
      hipLaunchKernelGGL(k_marginal_invchol_wave, dim3(std::min((M.nlist + NT / 64 - 1) / (NT / 64), 8 * h->sm_count)), dim3(NT),
                         (size_t)(NT / 64) * 64 * CH_LD * sizeof(double), st, M, cp);
    else hipLaunchKernelGGL(k_marginal_invchol, dim3(std::min(M.nlist, 8 * h->sm_count)), dim3(NT), lds, st, M, cp);
  }
  for (int g = g_lo; g < g_hi; ++g) {
    const LevelInfo &L = h->levels[g];
    if ((L.fast ? L.gown_n : L.own_n) == 0) continue;
    FactorArgs A;
    std::memset(&A, 0, sizeof(A));
    A.blks = h->d_blks.p; A.anc_idx = h->d_anc.p; A.list = h->d_lvl.p + L.first + L.own_lo; A.nlist = L.own_n;
    A.cx = h->d_cx.p; A.cy = h->d_cy.p; A.mv = h->d_mv.p; A.w_in = h->d_w.p; A.w_out = nullptr; A.z = nullptr;
    A.panels = h->d_panels[phys].p; A.logdet_c = h->d_logdet[phys].p; A.loglik_c = h->d_loglik[phys].p;
    A.errflag = errflag; A.maxP = L.maxP; A.maxM = L.maxM; A.maxMa = L.maxMa; A.SR = L.big_factor ? 4 : 8;
    {
      ProfScope ps(h, 0, g, 1, st);
      if (L.fast && h->factor_gen == 3 && L.q_nkx > 0) {
        QuadArgs F;
        std::memset(&F, 0, sizeof(F));
        F.blks = h->d_blks.p; F.anc_idx = h->d_anc.p; F.grps = h->d_grps.p + L.grp_first;
        F.quads = h->d_quads.p + L.quad_first + L.qown_lo; F.nquad = L.qown_n;
        F.cx = h->d_cx.p; F.cy = h->d_cy.p; F.mv = h->d_mv.p; F.w = h->d_w.p; F.panels = h->d_panels[phys].p;
        F.logdet_c = h->d_logdet[phys].p; F.loglik_c = h->d_loglik[phys].p; F.errflag = errflag; F.ldS = L.q_ldS;
        F.gdesc = h->d_gdesc.p + (size_t)L.grp_first * h->gd_stride; F.gd_stride = h->gd_stride;
        F.wave_chol = L.maxM <= 27 ? 1 : 0;
#define QLAUNCH(NU_, NKX_, NKT_)                                                                                               \
  do {                                                                                                                         \
    if (L.isref && F.wave_chol) hipLaunchKernelGGL((k_factor_quad<NU_, NKX_, NKT_, true, true>), dim3(L.qown_n), dim3(128 * NU_), L.lds_quad, st, F, cp); \
    else if (L.isref) hipLaunchKernelGGL((k_factor_quad<NU_, NKX_, NKT_, true, false>), dim3(L.qown_n), dim3(128 * NU_), L.lds_quad, st, F, cp); \
    else hipLaunchKernelGGL((k_factor_quad<NU_, NKX_, NKT_, false, true>), dim3(L.qown_n), dim3(128 * NU_), L.lds_quad, st, F, cp);         \
  } while (0)
        if (L.q_nkx == 32) QLAUNCH(4, 32, 8); else if (L.q_nkx == 38) QLAUNCH(4, 38, 10); else if (L.q_nkx == 44) QLAUNCH(4, 44, 11); else QLAUNCH(4, 50, 13);
#undef QLAUNCH
      } else if (L.fast) {
        FastArgs F;
        std::memset(&F, 0, sizeof(F));
        F.blks = h->d_blks.p; F.anc_idx = h->d_anc.p; F.grps = h->d_grps.p + L.grp_first + L.gown_lo; F.ngrp = L.gown_n;
        F.cx = h->d_cx.p; F.cy = h->d_cy.p; F.mv = h->d_mv.p; F.w = h->d_w.p; F.panels = h->d_panels[phys].p;
        F.logdet_c = h->d_logdet[phys].p; F.loglik_c = h->d_loglik[phys].p; F.errflag = errflag;
        F.Pm4 = L.Pm4; F.ldKV = L.ldKV; F.ldS = L.ldS; F.SRm = L.SRm; F.stage_dbl = L.stage_dbl;
        F.gdesc = h->d_gdesc.p + (size_t)(L.grp_first + L.gown_lo) * h->gd_stride; F.gd_stride = h->gd_stride;
        hipLaunchKernelGGL(k_factor_mfma, dim3(L.gown_n), dim3(NT), L.lds_fast, st, F, cp);
      } else if (L.bigmfma && h->factor_gen == 3 && L.lchain) {
        LcArgs C;
        std::memset(&C, 0, sizeof(C));
        C.blks = h->d_blks.p; C.anc_idx = h->d_anc.p; C.slabs = h->d_lcslabs.p + L.lc_first; C.nslab = L.lc_count;
        C.cx = h->d_cx.p; C.cy = h->d_cy.p; C.mv = h->d_mv.p; C.w_in = h->d_w.p; C.panels = h->d_panels[phys].p;
        C.rowtmp = h->d_lcrow.p; C.n_rows = h->n_all; C.errflag = errflag;
        C.errcode = L.lchain_ref ? 2 : 3;
        C.vscr = L.lchain_ref ? h->d_vscr.p : nullptr;
        if (L.lchain == 96) hipLaunchKernelGGL((k_factor_lchain<96>), dim3(L.lc_count), dim3(LC_NT), lc_dyn_doubles(96) * 8, st, C, cp);
        else hipLaunchKernelGGL((k_factor_lchain<136>), dim3(L.lc_count), dim3(LC_NT), lc_dyn_doubles(136) * 8, st, C, cp);
        if (L.lchain_ref) {   // the panels hold [ -r_j T_j | r_j ] per column: finish the blocks (Schur complement, factorisation, -Ri T in place)
          A.vscr = h->d_vscr.p; A.voff = h->d_rfvoff.p + L.rf_first; A.hvrow = h->d_lcrow.p;
          hipLaunchKernelGGL(k_factor_ref_finish, dim3(std::min(A.nlist, 2 * h->sm_count)), dim3(BM_NT), L.lds_ref_finish, st, A, cp);
        } else
        hipLaunchKernelGGL(k_lchain_scalars, dim3((A.nlist + 255) / 256), dim3(256), 0, st, h->d_blks.p, A.list, A.nlist, h->d_lcrow.p, h->n_all,
                           h->d_logdet[phys].p, h->d_loglik[phys].p);
      } else if (L.bigmfma && h->factor_gen == 3 && L.wide_count > 0) {
        WideArgs W;
        std::memset(&W, 0, sizeof(W));
        W.blks = h->d_blks.p; W.anc_idx = h->d_anc.p; W.list = h->d_lvl.p + L.first + L.own_lo; W.groups = h->d_wgrps.p + L.wide_first;
        W.ngroups = L.wide_count; W.cx = h->d_cx.p; W.cy = h->d_cy.p; W.mv = h->d_mv.p; W.w_in = h->d_w.p; W.panels = h->d_panels[phys].p;
        W.logdet_c = h->d_logdet[phys].p; W.loglik_c = h->d_loglik[phys].p; W.errflag = errflag; W.scratch = h->d_scratch.p;
        W.scratch_stride = h->scratch_stride; W.maxP = L.maxP; W.maxN = L.wide_maxN; W.maxM = L.maxM; W.maxMa = L.maxMa; W.ldS = L.bm_ldS;
        hipLaunchKernelGGL((k_factor_wide<WG_JT>), dim3(std::min(L.wide_count, h->sm_count)), dim3(WG_NT), L.lds_wide, st, W, cp);
      } else if (L.bigmfma && h->factor_gen == 3) {
        A.scratch = h->d_scratch.p; A.scratch_stride = h->scratch_stride; A.SR = L.bm_ldS;
        if (L.maxM <= 48) hipLaunchKernelGGL((k_factor_bigmfma<3, 5, 34>), dim3(std::min(A.nlist, h->sm_count)), dim3(BM_NT), L.lds_bigmfma, st, A, cp);
        else if (L.maxM <= 64) hipLaunchKernelGGL((k_factor_bigmfma<4, 5, 34>), dim3(std::min(A.nlist, h->sm_count)), dim3(BM_NT), L.lds_bigmfma, st, A, cp);
        else hipLaunchKernelGGL((k_factor_bigmfma<5, 3, 24>), dim3(std::min(A.nlist, h->sm_count)), dim3(BM_NT), L.lds_bigmfma, st, A, cp);
      } else if (L.big_factor) launch_factor<true, MODE_FACTOR>(h, L, A, cp, st);
      else launch_factor<false, MODE_FACTOR>(h, L, A, cp, st);
    }
    HCHK(h, hipGetLastError());
  }
  return ST_OK;
}


// Phase A of the top levels ahead of time, on the second stream: call before the sweep with the theta st_factor /
// st_factor_local will be given next for the same slot.  A no-op when the tree does not qualify (or SPAMTREE_ASYNC_TOP=0).
extern "C" int st_factor_ahead_levels(st_handle h) { return (h && h->async_top && !h->async_top_off) ? h->g_top : 0; }
// measurement: switch the ahead-of-time path off / on at run time (bench.py's per-level pass runs every level of phase A back
// to back on ONE stream; on the second stream, under the sweep, the top levels' launch times are not their own)
extern "C" int st_factor_ahead_enable(st_handle h, int enable) {
  if (!h) return ST_ERR_USAGE;
  h->async_top_off = !enable;
  return ST_OK;
}
extern "C" int st_factor_begin(st_handle h, int slot, const double *theta, int ntheta) {
  if (!h || !theta || slot < 0 || slot > 1) return ST_ERR_USAGE;
  if (!h->async_top || h->async_top_off) return ST_OK;
  HCHK(h, hipSetDevice(h->device));
  CovPar cp;
  int rc = make_covpar(h, theta, ntheta, &cp);
  if (rc) return rc;
  if (h->top_pending) HCHK(h, hipStreamWaitEvent(h->stream2, h->ev_top, 0));
  HCHK(h, hipEventRecord(h->ev_main, h->stream));          // everything issued so far (the previous iteration) comes first
  HCHK(h, hipStreamWaitEvent(h->stream2, h->ev_main, 0));
  const int init[2] = {INT_MAX, 0};
  HCHK(h, hipMemcpyAsync(h->d_err2.p, init, 2 * sizeof(int), hipMemcpyHostToDevice, h->stream2));
  const int phys = h->slot_map[slot];
  h->prof_suspend = true;   // not timed: the launches overlap the sweep on another stream
  rc = factor_launch(h, phys, cp, 0, h->g_top, h->stream2, h->d_err2.p);
  h->prof_suspend = false;
  if (rc) return rc;
  HCHK(h, hipEventRecord(h->ev_top, h->stream2));
  h->top_pending = true; h->top_phys = phys; h->top_theta.assign(theta, theta + ntheta);
  return ST_OK;
}
static int fix_top_comps(st_handle h, int phys);
static int run_stats(st_handle h, hipStream_t st);
// The beta / tausq statistics of the iteration need the sweep's w and the current XB only: when a proposal is about to be
// factorised they start on the second stream and run under phase A (the driver asks for them after the Metropolis step).
static int stats_begin(st_handle h) {
  if (!h->stream2 || h->stats_valid) return ST_OK;
  HCHK(h, hipEventRecord(h->ev_main, h->stream));           // w of the sweep is final at this point of the main stream
  HCHK(h, hipStreamWaitEvent(h->stream2, h->ev_main, 0));
  int rc = run_stats(h, h->stream2);
  if (rc) return rc;
  // the results travel to pinned host memory on the same stream: when the driver asks (after the Metropolis step) they are
  // there, and fetching them costs an event query instead of a copy + a synchronisation of the main stream
  const size_t nq = (size_t)h->p * h->q + h->q;
  h->stats_prefetched = nq <= 40;
  if (h->stats_prefetched) HCHK(h, hipMemcpyAsync(h->pin + 20, h->d_stats.p, nq * sizeof(double), hipMemcpyDeviceToHost, h->stream2));
  HCHK(h, hipEventRecord(h->ev_stats, h->stream2));
  h->stats_on_stream2 = true;
  return ST_OK;
}

extern "C" int st_factor_local(st_handle h, int slot, const double *theta, int ntheta) {
  if (!h || !theta || slot < 0 || slot > 1) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  CovPar cp;
  int rc = make_covpar(h, theta, ntheta, &cp);
  if (rc) return rc;
  h->theta[slot].assign(theta, theta + ntheta);
  if (slot == 0) { h->gram_valid = false; std::fill(h->s0_valid.begin(), h->s0_valid.end(), 0); }
  rc = reset_err(h);
  if (rc) return rc;
  const int phys = h->slot_map[slot];
  if (slot == 1) { rc = stats_begin(h); if (rc) return rc; }
  bool reuse = false;
  if (h->top_pending) {
    HCHK(h, hipStreamWaitEvent(h->stream, h->ev_top, 0));   // the top levels are done (or at least out of the way) before anything below
    reuse = h->top_phys == phys && (int)h->top_theta.size() == ntheta && std::equal(theta, theta + ntheta, h->top_theta.begin());
    h->top_pending = false;
  }
  if (!reuse) return factor_launch(h, phys, cp);
  rc = factor_launch(h, phys, cp, h->g_top);
  if (rc) return rc;
  rc = fix_top_comps(h, phys);
  if (rc) return rc;
  hipLaunchKernelGGL(k_merge_err, dim3(1), dim3(64), 0, h->stream, h->d_err.p, h->d_err2.p);
  HCHK(h, hipGetLastError());
  return ST_OK;
}

extern "C" int st_mg_pack_comps(st_handle h, int slot, void **dev_ptr, int64_t *len) {
  if (!h || slot < 0 || slot > 1) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  { const int rc0 = settle_top(h); if (rc0) return rc0; }
  const int phys = h->slot_map[slot], nb = (int)h->n_blocks;
  {
    ProfScope ps(h, 3);
    hipLaunchKernelGGL(k_pack_comps, dim3((std::max(nb, h->world) + NT - 1) / NT), dim3(NT), 0, h->stream, h->d_logdet[phys].p,
                       h->d_loglik[phys].p, h->d_blkmask.p, nb, h->d_err.p, h->rank, h->world, h->d_comm.p);
  }
  HCHK(h, hipGetLastError());
  if (dev_ptr) *dev_ptr = h->d_comm.p;
  if (len) *len = 2 * (int64_t)nb + h->world;
  return ST_OK;
}

// after the exchange of the packed components: deterministic sum + failure code agreed by all ranks
extern "C" int st_mg_finish(st_handle h, double *loglik) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  const int nb = (int)h->n_blocks;
  {
    ProfScope ps(h, 3);
    launch_sum2(h->stream, h->d_comm.p, h->d_comm.p + nb, nb, h->d_scalars.p + 8, h->d_scalars.p);
  }
  HCHK(h, hipGetLastError());
  double s2[2], errw[64];
  HCHK(h, hipMemcpyAsync(s2, h->d_scalars.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipMemcpyAsync(errw, h->d_comm.p + 2 * (size_t)nb, h->world * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  int best = INT_MAX;
  for (int r = 0; r < h->world; ++r) if (errw[r] > 0.5) best = std::min(best, (int)errw[r]);
  if (best != INT_MAX) return best & 15;
  if (loglik) *loglik = s2[0] + s2[1];   // loglik_w = logdetCi + sum(loglik_w_comps)  (:987-988, :815-816)
  return ST_OK;
}

#define NCHK(h, call)                                                                                        \
  do {                                                                                                        \
    ncclResult_t r_ = (call);                                                                                 \
    if (r_ != ncclSuccess) { (h)->err = std::string(#call) + ": " + ncclGetErrorString(r_); return ST_ERR_HIP; } \
  } while (0)

// pack -> RCCL all-reduce(sum) on the launch stream -> deterministic finish (native path of the multi-GPU protocol)
static int exchange_comps_and_finish(st_handle h, int slot, double *loglik) {
  void *ptr = nullptr;
  int64_t len = 0;
  int rc = st_mg_pack_comps(h, slot, &ptr, &len);
  if (rc) return rc;
  {
    ProfScope ps(h, 7);
    NCHK(h, ncclAllReduce(ptr, ptr, (size_t)len, ncclDouble, ncclSum, h->comm, h->stream));
  }
  return st_mg_finish(h, loglik);
}

extern "C" int st_comm_unique_id(void *out, int32_t cap) {
  if (!out || cap < (int32_t)sizeof(ncclUniqueId)) return ST_ERR_USAGE;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return ST_ERR_HIP;
  std::memcpy(out, &id, sizeof(id));
  return (int)sizeof(id);
}
extern "C" int st_comm_init(st_handle h, const void *unique_id) {
  if (!h || !unique_id) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  NCHK(h, ncclCommInitRank(&h->comm, h->world, id, h->rank));
  return ST_OK;
}

extern "C" int st_factor(st_handle h, int slot, const double *theta, int ntheta, double *loglik) {
  if (!h) return ST_ERR_USAGE;
  if (h->world > 1 || h->comm) {   // an attached communicator selects the exchange protocol even with one rank (tests)
    if (!h->comm) { h->err = "world > 1: call st_comm_init first, or use st_factor_local / st_mg_pack_comps / (all-reduce) / st_mg_finish"; return ST_ERR_USAGE; }
    int rc = st_factor_local(h, slot, theta, ntheta);
    if (rc) return rc;
    return exchange_comps_and_finish(h, slot, loglik);
  }
  int rc = st_factor_enqueue(h, slot, theta, ntheta);
  if (rc) return rc;
  return st_factor_finish(h, loglik);
}
// st_factor in two halves (one GPU): everything is ENQUEUED by the first -- the levels, the two sums, the copies of the sums and of
// the failure word to pinned memory, an event behind them -- and the second waits for that event and reads them.  In between the
// host may enqueue work that does not touch the proposal's slot: the C++ driver draws tausq and beta from the sweep's statistics
// (ready early in phase A: they run on the second stream) and uploads them, so that XB is current when phase A ends instead of
// two host round trips later.  With a communicator attached the first half enqueues nothing and the second does all of st_factor.
extern "C" int st_factor_is_async(st_handle h) { return (h && !(h->world > 1 || h->comm)) ? 1 : 0; }
extern "C" int st_factor_enqueue(st_handle h, int slot, const double *theta, int ntheta) {
  if (!h || !theta || slot < 0 || slot > 1) return ST_ERR_USAGE;
  if (h->factor_open) { h->err = "st_factor_enqueue: the previous one has not been finished"; return ST_ERR_USAGE; }
  if (h->world > 1 || h->comm) {
    h->factor_open = true; h->factor_open_slot = slot; h->top_theta_open.assign(theta, theta + ntheta);
    return ST_OK;
  }
  int rc = st_factor_local(h, slot, theta, ntheta);
  if (rc) return rc;
  // the failure word and the two sums come back in ONE synchronisation (the sums are meaningless after a failure)
  {
    ProfScope ps(h, 3);
    const int phys = h->slot_map[slot];
    launch_sum2(h->stream, h->d_logdet[phys].p, h->d_loglik[phys].p, (int)h->n_blocks, h->d_scalars.p + 8, h->d_scalars.p);
  }
  HCHK(h, hipGetLastError());
  HCHK(h, hipMemcpyAsync(h->pin + 12, h->d_scalars.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipMemcpyAsync(h->pin + 14, h->d_err.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipEventRecord(h->ev_factor, h->stream));
  h->factor_open = true; h->factor_open_slot = slot; h->top_theta_open.clear();
  return ST_OK;
}
extern "C" int st_factor_finish(st_handle h, double *loglik) {
  if (!h) return ST_ERR_USAGE;
  if (!h->factor_open) { h->err = "st_factor_finish without st_factor_enqueue"; return ST_ERR_USAGE; }
  h->factor_open = false;
  if (h->world > 1 || h->comm) {
    if (!h->comm) { h->err = "world > 1: call st_comm_init first, or use st_factor_local / st_mg_pack_comps / (all-reduce) / st_mg_finish"; return ST_ERR_USAGE; }
    int rc = st_factor_local(h, h->factor_open_slot, h->top_theta_open.data(), (int)h->top_theta_open.size());
    if (rc) return rc;
    return exchange_comps_and_finish(h, h->factor_open_slot, loglik);
  }
  HCHK(h, hipSetDevice(h->device));
  HCHK(h, hipEventSynchronize(h->ev_factor));
  const int e0 = ((const int *)(h->pin + 14))[0];
  if (e0 != INT_MAX) return e0 & 15;  // the reference's `return false` (:971-982); deeper levels hold unspecified values (Q5)
  if (loglik) *loglik = h->pin[12] + h->pin[13];   // loglik_w = logdetCi + sum(loglik_w_comps)  (:987-988)
  return ST_OK;
}

// st_sample_w followed by st_loglik_w(slot) with ONE synchronisation (the sweep's failure word travels with the sums).
static int gather_w_scatter(st_handle h);
extern "C" int st_sample_w_loglik(st_handle h, const double *z, uint64_t seed, uint32_t iter, int slot, double *loglik) {
  if (!h || slot < 0 || slot > 1) return ST_ERR_USAGE;
  if (h->world > 1 || h->comm) {   // an attached communicator selects the exchange protocol even with one rank (tests)
    if (!h->comm) {
      int rc = st_sample_w(h, z, seed, iter);   // reports the missing communicator
      if (rc) return rc;
      return st_loglik_w(h, slot, loglik);
    }
    // Fused exchange: the log-density of a rank's own blocks needs w of its own subtrees and of the replicated top only,
    // both current BEFORE the other ranks' rows arrive -- so phase C runs ahead of the exchange of w, and w and the
    // log-density components travel in ONE grouped RCCL call, followed by ONE host synchronisation.
    int rc = st_sample_w_local(h, z, seed, iter);
    if (rc) return rc;
    if (h->top_len > 0) {
      ProfScope ps(h, 7);
      NCHK(h, ncclAllReduce(h->d_acc.p + h->top_off, h->d_acc.p + h->top_off, (size_t)h->top_len, ncclDouble, ncclSum, h->comm, h->stream));
    }
    rc = st_sample_w_top(h);
    if (rc) return rc;
    void *pw = nullptr, *pr = nullptr, *pc = nullptr;
    int64_t lw = 0, lc = 0;
    rc = st_mg_gather_w_pack(h, &pw, &pr, &lw);   // own rows of w + this rank's failure word of the sweep
    if (rc) return rc;
    rc = st_loglik_local(h, slot);          // resets the failure word after the pack above (stream order)
    if (rc) return rc;
    rc = st_mg_pack_comps(h, slot, &pc, &lc);
    if (rc) return rc;
    {
      ProfScope ps(h, 7);
      NCHK(h, ncclGroupStart());
      NCHK(h, ncclAllGather(pw, pr, (size_t)lw, ncclDouble, h->comm, h->stream));
      NCHK(h, ncclAllReduce(pc, pc, (size_t)lc, ncclDouble, ncclSum, h->comm, h->stream));
      NCHK(h, ncclGroupEnd());
    }
    rc = gather_w_scatter(h);
    if (rc) return rc;
    invalidate_stats(h);
    const int nb = (int)h->n_blocks;
    {
      ProfScope ps(h, 3);
      launch_sum2(h->stream, h->d_comm.p, h->d_comm.p + nb, nb, h->d_scalars.p + 8, h->d_scalars.p);
    }
    HCHK(h, hipGetLastError());
    double s2[2], errw[64], errc[64];
    HCHK(h, hipMemcpyAsync(errw, h->d_gerr.p, h->world * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HCHK(h, hipMemcpyAsync(s2, h->d_scalars.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HCHK(h, hipMemcpyAsync(errc, h->d_comm.p + 2 * (size_t)nb, h->world * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HCHK(h, hipStreamSynchronize(h->stream));
    int best = INT_MAX;
    for (int r = 0; r < h->world; ++r) if (errw[r] > 0.5) best = std::min(best, (int)errw[r]);
    if (best != INT_MAX) return best & 15;   // 10 / 11: the reference stops with "Error at gibbs_sample_w" (:1215-1217)
    for (int r = 0; r < h->world; ++r) if (errc[r] > 0.5) best = std::min(best, (int)errc[r]);
    if (best != INT_MAX) return best & 15;
    if (loglik) *loglik = s2[0] + s2[1];
    return ST_OK;
  }
  int rc = st_sample_w_local(h, z, seed, iter);
  if (rc) return rc;
  rc = st_sample_w_top(h);
  if (rc) return rc;
  HCHK(h, hipMemcpyAsync(h->pin + 2, h->d_err.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  rc = st_loglik_local(h, slot);   // resets the failure word after the copy above (stream order)
  if (rc) return rc;
  {
    ProfScope ps(h, 3);
    const int phys = h->slot_map[slot];
    launch_sum2(h->stream, h->d_logdet[phys].p, h->d_loglik[phys].p, (int)h->n_blocks, h->d_scalars.p + 8, h->d_scalars.p);
  }
  HCHK(h, hipGetLastError());
  HCHK(h, hipMemcpyAsync(h->pin, h->d_scalars.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  const int e0 = ((const int *)(h->pin + 2))[0];
  if (e0 != INT_MAX) return e0 & 15;  // 10 / 11: the reference stops with "Error at gibbs_sample_w" (:1215-1217)
  if (loglik) *loglik = h->pin[0] + h->pin[1];
  return ST_OK;
}

// The same pair WITHOUT the host synchronisation: the log-density of the sweep's w is not needed on the host before the
// Metropolis step, i.e. after the proposal's factorisation -- whose own synchronisation then brings it along.  One round trip
// and one idle gap of the GPU less per iteration (the driver enqueues phase A right behind phase C).  _end returns what the
// synchronous call would have returned (failure code of the sweep / the log-density); it synchronises only if nobody has yet.
// With a communicator attached (multi-GPU) _begin simply runs the synchronous protocol and _end hands its result over.
extern "C" int st_sample_w_loglik_begin(st_handle h, const double *z, uint64_t seed, uint32_t iter, int slot) {
  if (!h || slot < 0 || slot > 1) return ST_ERR_USAGE;
  if (h->c_pending) return ST_ERR_USAGE;
  if (h->world > 1 || h->comm) {
    h->c_ll = 0.0;
    h->c_rc = st_sample_w_loglik(h, z, seed, iter, slot, &h->c_ll);
    if (h->c_rc < 0) return h->c_rc;   // a HIP / RCCL / usage error: nothing is pending, the caller must not call _end
    h->c_pending = true;
    return ST_OK;
  }
  int rc = st_sample_w_local(h, z, seed, iter);
  if (rc) return rc;
  rc = st_sample_w_top(h);
  if (rc) return rc;
  HCHK(h, hipMemcpyAsync(h->pin + 10, h->d_err.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  rc = st_loglik_local(h, slot);   // resets the failure word after the copy above (stream order)
  if (rc) return rc;
  {
    ProfScope ps(h, 3);
    const int phys = h->slot_map[slot];
    launch_sum2(h->stream, h->d_logdet[phys].p, h->d_loglik[phys].p, (int)h->n_blocks, h->d_scalars.p + 8, h->d_scalars.p);
  }
  HCHK(h, hipGetLastError());
  HCHK(h, hipMemcpyAsync(h->pin + 8, h->d_scalars.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  h->c_rc = INT_MIN;   // not known yet
  h->c_pending = true;
  return ST_OK;
}
extern "C" int st_sample_w_loglik_end(st_handle h, double *loglik) {
  if (!h || !h->c_pending) return ST_ERR_USAGE;
  h->c_pending = false;
  if (h->c_rc != INT_MIN) {   // the synchronous protocol ran in _begin
    if (h->c_rc == ST_OK && loglik) *loglik = h->c_ll;
    return h->c_rc;
  }
  HCHK(h, hipStreamSynchronize(h->stream));   // returns at once when a later call (st_factor) has synchronised already
  const int e0 = ((const int *)(h->pin + 10))[0];
  if (e0 != INT_MAX) return e0 & 15;   // 10 / 11: "Error at gibbs_sample_w"
  if (loglik) *loglik = h->pin[8] + h->pin[9];
  return ST_OK;
}

static int gen_or_upload_z(st_handle h, const double *z, uint64_t seed, uint32_t iter, unsigned stream_id, double *dst) {
  if (z) return upload_rows(h, z, dst);
  {
    ProfScope ps(h, 5);
    const int grid = (int)((h->n_all + NT - 1) / NT);
    hipLaunchKernelGGL(k_normals, dim3(grid), dim3(NT), 0, h->stream, dst, h->d_dev2model.p, h->n_all, iter, stream_id,
                       (unsigned long long)seed);
  }
  HCHK(h, hipGetLastError());
  return ST_OK;
}

static int sample_launch(st_handle h, int g_hi, int g_lo) {   // levels g_hi-1 ... g_lo
  const int phys = h->slot_map[0];
  for (int g = g_hi - 1; g >= g_lo; --g) {
    const LevelInfo &L = h->levels[g];
    if ((L.fast ? L.gown_n : L.own_n) == 0) continue;
    SampleArgs A;
    std::memset(&A, 0, sizeof(A));
    A.blks = h->d_blks.p; A.anc_idx = h->d_anc.p; A.dch_idx = h->d_dch.p; A.list = h->d_lvl.p + L.first + L.own_lo; A.nlist = L.own_n;
    A.panels = h->d_panels[phys].p; A.w = h->d_w.p; A.y = h->d_y.p; A.xb = h->d_xb.p; A.z = h->d_z.p; A.mv = h->d_mv.p;
    A.obs = h->d_obs.p; A.acc = h->d_acc.p; A.errflag = h->d_err.p; A.maxP = L.maxP; A.maxM = L.maxM; A.maxLd = L.maxLd;
    for (int j = 0; j < QMAX; ++j) A.tausq_inv[j] = h->tausq_inv[j];
    A.do_gram = (h->gram_valid && h->cache_gram) ? 0 : 1;
    A.no_fwd = h->limited ? 1 : 0;
    A.lds_sq = (L.big_sample && L.sample_sq) ? 2 : 0;
    A.s0 = h->d_s0.p; A.s0off = h->d_s0off.p;
    {
      ProfScope ps(h, 1, h->n_actual_groups + g);   // per-level slots of phase B follow those of phase A
      if (L.fast) {
        SampleFastArgs F;
        std::memset(&F, 0, sizeof(F));
        F.do_gram = A.do_gram; F.no_fwd = A.no_fwd;
        F.blks = h->d_blks.p; F.anc_idx = h->d_anc.p; F.dch_idx = h->d_dch.p; F.grps = h->d_grps.p + L.grp_first + L.gown_lo; F.ngrp = L.gown_n;
        F.panels = h->d_panels[phys].p; F.w = h->d_w.p; F.y = h->d_y.p; F.xb = h->d_xb.p; F.z = h->d_z.p; F.mv = h->d_mv.p;
        F.acc = h->d_acc.p; F.errflag = h->d_err.p; F.ldN = L.ldN; F.Mr4 = L.Mr4; F.Mrows = L.Mrows; F.maxP = L.maxP; F.av_dbl = L.av_dbl;
        F.gdesc = h->d_gdesc.p + (size_t)(L.grp_first + L.gown_lo) * h->gd_stride; F.gd_stride = h->gd_stride;
        for (int j = 0; j < QMAX; ++j) F.tausq_inv[j] = h->tausq_inv[j];
        const bool lean_ok = h->sample_lean != 0 && !(!L.isref && L.maxP > 255);
        F.gdesc_all = h->d_gdesc.p;
        // does level `gq` take k_gram + the lean kernels on a rebuild sweep?  (the same test for the level itself, below)
        auto splits = [&](int gq) {
          const LevelInfo &Lq = h->levels[gq];
          const bool lean_q = h->sample_lean != 0 && !(!Lq.isref && Lq.maxP > 255);
          return Lq.fast && lean_q && h->split_gram && (h->split_gram == 2 || (Lq.isref && Lq.gown_n >= 32 * h->sm_count));
        };
        // the leaf level of a rebuild sweep writes NO Gram parts when its parents form them from the leaf panels (k_gram_direct)
        const bool direct_parent = F.do_gram && h->gram_direct_level == g && splits(g) && h->levels[g + 1].gown_n > 0;
        if (F.do_gram && lean_ok && g == h->gram_direct_level + 1 && h->gram_direct_level >= 0 && splits(g - 1) && h->levels[g - 1].gown_n > 0) F.do_gram = 0;
        // the theta-only Gram parts on their own (k_gram), then the lean sweep kernels: pays on big reference levels (n = 1e6,
        // level 7: 0.84 -> 0.72 ms averaged over a run's sweeps), loses on leaf levels and on smaller reference levels, where
        // the staged Gram of k_sample_mfma is cheaper (SPAMTREE_SPLIT_GRAM=2: every level; records identical either way)
        if (F.do_gram && lean_ok && h->split_gram && (h->split_gram == 2 || (L.isref && L.gown_n >= 32 * h->sm_count))) {
          if (direct_parent) hipLaunchKernelGGL(k_gram_direct, dim3(L.gown_n), dim3(NT), 0, h->stream, F);
          else hipLaunchKernelGGL(k_gram, dim3(L.gown_n), dim3(NT), 0, h->stream, F);
          F.do_gram = 0;
        }
        // reference levels on the lean kernels: the theta-only part of the posterior precision (Ri' Ri + the children's Gram parts)
        // is stored by the first such sweep after the records were rebuilt and loaded by the following ones (same bits either way)
        if (!(F.do_gram || !lean_ok) && L.isref) {
          F.s0 = h->d_s0.p; F.s0off = h->d_s0off.p;
          F.s0_mode = !h->cache_gram ? 0 : (h->s0_valid[g] ? 2 : 1);
          if (h->cache_gram) h->s0_valid[g] = 1;
        }
        if (F.do_gram || !lean_ok) hipLaunchKernelGGL(k_sample_mfma, dim3(L.gown_n), dim3(NT), L.lds_sfast, h->stream, F);
        else if (!L.isref) {
          // segment-aligned lanes where the level's chains have at most 12 ancestors of at most 32 rows (SPAMTREE_LEAF_SEG=0: the
          // column-aligned kernel)
          const int need = (L.maxJ + 1) / 2;
          if (h->leaf_seg && need <= 4 && L.maxMa <= 32) hipLaunchKernelGGL((k_sample_leaf_seg<4>), dim3(L.gown_n), dim3(NT), ((size_t)4 * 64 * 4 + 3 * 32) * 8, h->stream, F);
          else if (h->leaf_seg && need <= 6 && L.maxMa <= 32) hipLaunchKernelGGL((k_sample_leaf_seg<6>), dim3(L.gown_n), dim3(NT), ((size_t)4 * 64 * 6 + 3 * 32) * 8, h->stream, F);
          else hipLaunchKernelGGL(k_sample_leaf, dim3(L.gown_n), dim3(NT), ((size_t)L.maxP + 32 + 4 * 256 + 3 * 32) * 8, h->stream, F);
        }
        else if (h->sample_wave && L.maxM <= 27 && (h->sample_wave == 2 || L.gown_n >= 8 * h->sm_count)) {   // one block per wave: faster on a level
          // that keeps every CU busy for several rounds (n = 1e6 after the row-wise panel pass: level 7 0.44 -> 0.35 ms, level 6 -- 4096 blocks --
          // 0.151 -> 0.117), a wash at 1024 blocks (0.050 -> 0.046 there, 0.034 -> 0.039 at config #5), slower on latency-bound small levels (256 blocks: 0.029 -> 0.040): gd | wv | seg | tv, ev | Ri, per wave
          const size_t per = (((size_t)h->gd_stride + L.maxP + 32 + L.av_dbl + 64 + (size_t)std::max(L.maxM, 1) * CH_LD + 1) & ~(size_t)1);
          F.ldN = (int)per; F.Mrows = L.maxM;
          hipLaunchKernelGGL(k_sample_wave, dim3((L.gown_n + NT / 64 - 1) / (NT / 64)), dim3(NT), per * 8 * (NT / 64), h->stream, F);
        } else {
          // levels that do not fill the chip (fewer groups than 2 x CUs x the five workgroups the occupancy variant fits): the
          // latency variant -- every descriptor-only load of a block in one round trip (SPAMTREE_SAMPLE_LAT=0: never; identical draws)
          static const bool lat_on = !(getenv("SPAMTREE_SAMPLE_LAT") && getenv("SPAMTREE_SAMPLE_LAT")[0] == '0');
          F.av_dbl = L.av_dbl + 224;
          if (lat_on && L.gown_n <= 2 * h->sm_count) hipLaunchKernelGGL((k_sample_lean<true>), dim3(L.gown_n), dim3(NT), L.lds_slean, h->stream, F);
          else hipLaunchKernelGGL((k_sample_lean<false>), dim3(L.gown_n), dim3(NT), L.lds_slean, h->stream, F);
        }
      } else {
        if (A.do_gram && L.big_sample && h->gram_big) {
          // the theta-only parts first, on the matrix cores (k_gram_big); the sweep kernel then takes its cached branch
          GramBigArgs Gb;
          std::memset(&Gb, 0, sizeof(Gb));
          Gb.blks = h->d_blks.p; Gb.anc_idx = h->d_anc.p; Gb.dch_idx = h->d_dch.p; Gb.list = A.list; Gb.nlist = A.nlist;
          Gb.panels = A.panels; Gb.acc = A.acc; Gb.s0 = A.s0; Gb.s0off = A.s0off; Gb.no_fwd = A.no_fwd;
          // both triangles unless every level of the tree is read by k_gram_big / the generic sweep's cached branch (tiles it >= jt /
          // the lower triangle only)
          bool all_big = true;
          for (const auto &L2 : h->levels) all_big = all_big && (L2.count == 0 || L2.big_sample);
          Gb.mirror = all_big ? 0 : 1;
          hipLaunchKernelGGL(k_gram_big, dim3(A.nlist), dim3(NT), 0, h->stream, Gb);
          A.do_gram = 0;
        }
        if (L.big_sample) {
          A.scratch = h->d_scratch.p; A.scratch_stride = h->scratch_stride;
          if (L.isref) hipLaunchKernelGGL((k_sample<true, false>), dim3(std::min(L.own_n, h->scratch_wgs)), dim3(NT), L.lds_sample, h->stream, A);
          else if (h->leaf_wide && !A.do_gram && L.maxM <= 64 && L.maxMa <= 96 && L.maxJ <= 8)   // one coalesced pass, segment-aligned lanes
            hipLaunchKernelGGL(k_sample_leaf_wide, dim3(std::min(L.own_n, 16 * h->sm_count)), dim3(NT), ((size_t)4 * 64 * 12 + 3 * 64) * 8, h->stream, A);
          else hipLaunchKernelGGL((k_sample<true, true>), dim3(std::min(L.own_n, h->scratch_wgs)), dim3(NT), L.lds_sample, h->stream, A);
        } else {
          hipLaunchKernelGGL((k_sample<false>), dim3(L.own_n), dim3(NT), L.lds_sample, h->stream, A);
        }
      }
    }
    HCHK(h, hipGetLastError());
  }
  return ST_OK;
}

// levels below the cut (this rank's subtrees); then the cut level's message records of the other ranks are zeroed so
// that an all-reduce(sum) over st_mg_top_region() completes them
extern "C" int st_sample_w_local(st_handle h, const double *z, uint64_t seed, uint32_t iter) {
  if (!h) return ST_ERR_USAGE;
  invalidate_stats(h);
  HCHK(h, hipSetDevice(h->device));
  int rc = gen_or_upload_z(h, z, seed, iter, 0u, h->d_z.p);
  if (rc) return rc;
  h->z_valid = true;
  rc = reset_err(h);
  if (rc) return rc;
  rc = sample_launch(h, h->n_actual_groups, std::min(h->cut, h->n_actual_groups));
  if (rc) return rc;
  for (auto &zr : h->top_zero) HCHK(h, hipMemsetAsync(h->d_acc.p + zr.first, 0, (size_t)zr.second * sizeof(double), h->stream));
  return ST_OK;
}
extern "C" int st_mg_top_region(st_handle h, void **dev_ptr, int64_t *len) {
  if (!h) return ST_ERR_USAGE;
  if (dev_ptr) *dev_ptr = h->d_acc.p + h->top_off;
  if (len) *len = h->top_len;
  return ST_OK;
}
extern "C" int st_sample_w_top(st_handle h) {   // the replicated levels above the cut
  if (!h) return ST_ERR_USAGE;
  invalidate_stats(h);
  HCHK(h, hipSetDevice(h->device));
  const int rc = sample_launch(h, std::min(h->cut, h->n_actual_groups), 0);
  if (rc == ST_OK) h->gram_valid = true;   // every record now carries the Gram sums of the accepted theta
  return rc;
}
extern "C" int st_mg_pack_w(st_handle h, void **dev_ptr, int64_t *len) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  const long long nn = std::max<long long>(h->n_all, h->world);
  hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((nn + NT - 1) / NT)), dim3(NT), 0, h->stream, h->d_w.p, h->d_rowmask.p, h->n_all, h->d_err.p,
                     h->rank, h->world, h->d_tmp_n.p);
  HCHK(h, hipGetLastError());
  if (dev_ptr) *dev_ptr = h->d_tmp_n.p;
  if (len) *len = h->n_all + h->world;
  return ST_OK;
}
extern "C" int st_mg_unpack_w(st_handle h) {
  if (!h) return ST_ERR_USAGE;
  invalidate_stats(h);
  HCHK(h, hipSetDevice(h->device));
  double errw[64];
  HCHK(h, hipMemcpyAsync(h->d_w.p, h->d_tmp_n.p, (size_t)h->n_all * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  HCHK(h, hipMemcpyAsync(errw, h->d_tmp_n.p + h->n_all, h->world * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  int best = INT_MAX;
  for (int r = 0; r < h->world; ++r) if (errw[r] > 0.5) best = std::min(best, (int)errw[r]);
  return best == INT_MAX ? ST_OK : (best & 15);
}

// all-gather form (half the traffic of the all-reduce of n doubles): pack -> all-gather(recv, count per rank) -> unpack
extern "C" int st_mg_gather_w_pack(st_handle h, void **send_ptr, void **recv_ptr, int64_t *count_per_rank) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  const int cnt = h->gather_cnt;
  double *mine = h->d_gather.p + (size_t)h->rank * cnt;
  hipLaunchKernelGGL(k_gather_pack, dim3((cnt + NT - 1) / NT), dim3(NT), 0, h->stream, h->d_w.p, h->d_gidx.p + (size_t)h->rank * cnt, cnt,
                     h->d_err.p, mine);
  HCHK(h, hipGetLastError());
  if (send_ptr) *send_ptr = mine;
  if (recv_ptr) *recv_ptr = h->d_gather.p;
  if (count_per_rank) *count_per_rank = cnt;
  return ST_OK;
}
static int gather_w_scatter(st_handle h) {   // launches only: rows into w, failure words into d_gerr
  const long long total = (long long)h->world * h->gather_cnt;
  hipLaunchKernelGGL(k_gather_unpack, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, h->stream, h->d_gather.p, h->d_gidx.p, h->gather_cnt,
                     total, h->d_w.p, h->d_gerr.p);
  HCHK(h, hipGetLastError());
  return ST_OK;
}
extern "C" int st_mg_gather_w_unpack(st_handle h) {
  if (!h) return ST_ERR_USAGE;
  invalidate_stats(h);
  HCHK(h, hipSetDevice(h->device));
  int rc = gather_w_scatter(h);
  if (rc) return rc;
  double errw[64];
  HCHK(h, hipMemcpyAsync(errw, h->d_gerr.p, h->world * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  int best = INT_MAX;
  for (int r = 0; r < h->world; ++r) if (errw[r] > 0.5) best = std::min(best, (int)errw[r]);
  return best == INT_MAX ? ST_OK : (best & 15);
}

extern "C" int st_sample_w(st_handle h, const double *z, uint64_t seed, uint32_t iter) {
  if (!h) return ST_ERR_USAGE;
  if (h->world > 1 || h->comm) {   // an attached communicator selects the exchange protocol even with one rank (tests)
    if (!h->comm) { h->err = "world > 1: call st_comm_init first, or use st_sample_w_local / st_mg_top_region / st_sample_w_top / st_mg_pack_w / st_mg_unpack_w"; return ST_ERR_USAGE; }
    int rc = st_sample_w_local(h, z, seed, iter);
    if (rc) return rc;
    if (h->top_len > 0) {
      ProfScope ps(h, 7);
      NCHK(h, ncclAllReduce(h->d_acc.p + h->top_off, h->d_acc.p + h->top_off, (size_t)h->top_len, ncclDouble, ncclSum, h->comm, h->stream));
    }
    rc = st_sample_w_top(h);
    if (rc) return rc;
    void *snd = nullptr, *rcv = nullptr;
    int64_t cnt = 0;
    rc = st_mg_gather_w_pack(h, &snd, &rcv, &cnt);
    if (rc) return rc;
    {
      ProfScope ps(h, 7);
      NCHK(h, ncclAllGather(snd, rcv, (size_t)cnt, ncclDouble, h->comm, h->stream));
    }
    return st_mg_gather_w_unpack(h);
  }
  int rc = st_sample_w_local(h, z, seed, iter);
  if (rc) return rc;
  rc = st_sample_w_top(h);
  if (rc) return rc;
  int code = 0;
  rc = read_err(h, &code);
  if (rc) return rc;
  return code;  // 10 / 11: the reference stops with "Error at gibbs_sample_w" (:1215-1217)
}

extern "C" int st_loglik_local(st_handle h, int slot) {
  if (!h || slot < 0 || slot > 1) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  if (slot == 1) { const int rc0 = settle_top(h); if (rc0) return rc0; }
  const int phys = h->slot_map[slot];
  int maxP = 0, maxM = 0;
  for (auto &L : h->levels) { maxP = std::max(maxP, L.maxP); maxM = std::max(maxM, L.maxM); }
  const size_t lds = lds_loglik_bytes(maxP, maxM);
  LoglikArgs A;
  A.blks = h->d_blks.p; A.anc_idx = h->d_anc.p; A.list = h->d_ownslow.p; A.nlist = (int)h->own_obs_slow.size();
  A.panels = h->d_panels[phys].p; A.w = h->d_w.p; A.loglik_c = h->d_loglik[phys].p; A.maxP = maxP; A.maxM = maxM;
  {
    ProfScope ps(h, 2);
    if (A.nlist > 0) hipLaunchKernelGGL(k_loglik, dim3(A.nlist), dim3(NT), lds, h->stream, A);
    if (!h->own_grp_list.empty()) {
      LoglikGrpArgs Gr;
      Gr.blks = h->d_blks.p; Gr.anc_idx = h->d_anc.p; Gr.grps = h->d_grps.p; Gr.list = h->d_owngrp.p; Gr.nlist = (int)h->own_grp_list.size();
      Gr.panels = h->d_panels[phys].p; Gr.w = h->d_w.p; Gr.loglik_c = h->d_loglik[phys].p; Gr.maxP = maxP;
      Gr.gdesc = h->d_gdesc.p; Gr.gd_stride = h->gd_stride;
      hipLaunchKernelGGL(k_loglik_grp, dim3(Gr.nlist), dim3(NT), (size_t)(maxP + 32) * sizeof(double), h->stream, Gr);
    }
  }
  HCHK(h, hipGetLastError());
  HCHK(h, reset_err(h) == ST_OK ? hipSuccess : hipErrorUnknown);
  return ST_OK;
}
// the quadratic forms of the top blocks with the CURRENT w (st_factor_begin ran them with the w of the sweep's start)
static int fix_top_comps(st_handle h, int phys) {
  if (h->n_toplist == 0) return ST_OK;
  int maxP = 0, maxM = 0;
  for (int g = 0; g < h->g_top; ++g) { maxP = std::max(maxP, h->levels[g].maxP); maxM = std::max(maxM, h->levels[g].maxM); }
  LoglikArgs A;
  A.blks = h->d_blks.p; A.anc_idx = h->d_anc.p; A.list = h->d_toplist.p; A.nlist = h->n_toplist;
  A.panels = h->d_panels[phys].p; A.w = h->d_w.p; A.loglik_c = h->d_loglik[phys].p; A.maxP = maxP; A.maxM = maxM;
  {
    ProfScope ps(h, 0, h->g_top > 0 ? h->g_top - 1 : 0);
    hipLaunchKernelGGL(k_loglik, dim3(A.nlist), dim3(NT), lds_loglik_bytes(maxP, maxM), h->stream, A);
  }
  HCHK(h, hipGetLastError());
  return ST_OK;
}
extern "C" int st_loglik_w(st_handle h, int slot, double *loglik) {
  if (!h) return ST_ERR_USAGE;
  if (h->world > 1 || h->comm) {   // an attached communicator selects the exchange protocol even with one rank (tests)
    if (!h->comm) { h->err = "world > 1: call st_comm_init first, or use st_loglik_local / st_mg_pack_comps / (all-reduce) / st_mg_finish"; return ST_ERR_USAGE; }
    int rc = st_loglik_local(h, slot);
    if (rc) return rc;
    return exchange_comps_and_finish(h, slot, loglik);
  }
  int rc = st_loglik_local(h, slot);
  if (rc) return rc;
  double ll = 0.0;
  rc = reduce_loglik(h, h->slot_map[slot], &ll);
  if (rc) return rc;
  if (loglik) *loglik = ll;
  return ST_OK;
}

extern "C" int st_predict(st_handle h, int theta_changed) {
  (void)theta_changed;  // H of a prediction block is rebuilt from the ancestor chain every call: same values as the cache
  if (!h) return ST_ERR_USAGE;
  if (h->pred_list.empty()) return ST_OK;
  invalidate_stats(h);
  if (!h->z_valid) { h->err = "st_predict needs the normals of a preceding st_sample_w (spamtree_model.cpp:1325)"; return ST_ERR_USAGE; }
  if (h->theta[0].empty()) { h->err = "st_predict before st_factor(slot 0)"; return ST_ERR_USAGE; }
  HCHK(h, hipSetDevice(h->device));
  CovPar cp;
  int rc = make_covpar(h, h->theta[0].data(), (int)h->theta[0].size(), &cp);
  if (rc) return rc;
  const LevelInfo &L = h->pred_info;
  if (h->pred_nkx > 0 && h->pred_quad_count > 0) {
    QuadArgs F;
    std::memset(&F, 0, sizeof(F));
    F.blks = h->d_blks.p; F.anc_idx = h->d_anc.p; F.grps = h->d_grps.p + h->pred_grp_first;
    F.quads = h->d_quads.p + h->pred_quad_first; F.nquad = h->pred_quad_count;
    F.cx = h->d_cx.p; F.cy = h->d_cy.p; F.mv = h->d_mv.p; F.w = h->d_w.p; F.panels = h->d_panels[h->slot_map[0]].p;
    F.logdet_c = nullptr; F.loglik_c = nullptr; F.errflag = h->d_err.p; F.ldS = quad_lds_stride(h->pred_nkx);
    F.gdesc = h->d_gdesc.p + (size_t)h->pred_grp_first * h->gd_stride; F.gd_stride = h->gd_stride;
    F.wave_chol = 1; F.predict = 1; F.z = h->d_z.p; F.w_out = h->d_w.p;
    {
      ProfScope ps(h, 6);
      const dim3 grid(h->pred_quad_count), blk(128 * 4);
      if (h->pred_nkx == 32) hipLaunchKernelGGL((k_factor_quad<4, 32, 8, false, true>), grid, blk, h->pred_lds, h->stream, F, cp);
      else if (h->pred_nkx == 38) hipLaunchKernelGGL((k_factor_quad<4, 38, 10, false, true>), grid, blk, h->pred_lds, h->stream, F, cp);
      else if (h->pred_nkx == 44) hipLaunchKernelGGL((k_factor_quad<4, 44, 11, false, true>), grid, blk, h->pred_lds, h->stream, F, cp);
      else hipLaunchKernelGGL((k_factor_quad<4, 50, 13, false, true>), grid, blk, h->pred_lds, h->stream, F, cp);
    }
    HCHK(h, hipGetLastError());
    HCHK(h, hipStreamSynchronize(h->stream));
    return ST_OK;
  }
  FactorArgs A;
  std::memset(&A, 0, sizeof(A));
  A.blks = h->d_blks.p; A.anc_idx = h->d_anc.p; A.list = h->d_pred.p; A.nlist = (int)h->pred_list.size();
  A.cx = h->d_cx.p; A.cy = h->d_cy.p; A.mv = h->d_mv.p; A.w_in = h->d_w.p; A.w_out = h->d_w.p; A.z = h->d_z.p;
  A.panels = h->d_panels[h->slot_map[0]].p; A.logdet_c = nullptr; A.loglik_c = nullptr; A.errflag = h->d_err.p;
  A.maxP = L.maxP; A.maxM = L.maxM; A.maxMa = L.maxMa; A.SR = L.big_factor ? 4 : 8;
  {
    ProfScope ps(h, 6);
    if (L.big_factor) launch_factor<true, MODE_PREDICT>(h, L, A, cp);
    else launch_factor<false, MODE_PREDICT>(h, L, A, cp);
  }
  HCHK(h, hipGetLastError());
  HCHK(h, hipStreamSynchronize(h->stream));
  return ST_OK;
}

static int run_stats(st_handle h, hipStream_t st = nullptr) {
  const int nq = h->p * h->q + h->q;
  if (h->stats_valid) return ST_OK;   // w and XB unchanged since the last reduction: both statistics are still current
  if (!st) st = h->stream;
  {
    ProfScope ps(h, 4, -1, 1, st);
    hipLaunchKernelGGL(k_stats, dim3(STATS_WG), dim3(NT), 0, st, h->d_X.p, h->d_y.p, h->d_w.p, h->d_xb.p, h->d_mv.p, h->d_obs.p,
                       h->d_partner.p, h->n_all, h->p, h->q, h->d_partial.p);
    hipLaunchKernelGGL(k_stats_final, dim3(nq), dim3(NT), 0, st, h->d_partial.p, STATS_WG, nq, h->d_stats.p);
  }
  HCHK(h, hipGetLastError());
  h->stats_valid = true;
  return ST_OK;
}
// both statistics travel to the host together; a second request for the same (w, XB) is served from the host copy
static int fetch_stats(st_handle h) {
  if (h->stats_valid && h->host_stats_valid) return ST_OK;
  int rc = run_stats(h);
  if (rc) return rc;
  const size_t nq = (size_t)h->p * h->q + h->q;
  h->host_stats.resize(nq);
  if (h->stats_on_stream2) {   // started under phase A (stats_begin)
    if (h->stats_prefetched) {   // ... and already copied to pinned memory behind them on the second stream
      HCHK(h, hipEventSynchronize(h->ev_stats));
      HCHK(h, hipStreamWaitEvent(h->stream, h->ev_stats, 0));   // later work on the main stream stays ordered behind the reduction
      h->stats_on_stream2 = false;
      for (size_t i = 0; i < nq; ++i) h->host_stats[i] = h->pin[20 + i];
      h->host_stats_valid = true;
      return ST_OK;
    }
    HCHK(h, hipStreamWaitEvent(h->stream, h->ev_stats, 0));   // its results before the copy
    h->stats_on_stream2 = false;
  }
  HCHK(h, hipMemcpyAsync(h->host_stats.data(), h->d_stats.p, nq * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  h->host_stats_valid = true;
  return ST_OK;
}
extern "C" int st_beta_stats(st_handle h, double *xty) {
  if (!h || !xty) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  int rc = fetch_stats(h);
  if (rc) return rc;
  for (int i = 0; i < h->p * h->q; ++i) xty[i] = h->host_stats[i];
  return ST_OK;
}
extern "C" int st_tausq_stats(st_handle h, double *ssq, int64_t *n_obs_by_q) {
  if (!h || !ssq) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  int rc = fetch_stats(h);
  if (rc) return rc;
  for (int j = 0; j < h->q; ++j) ssq[j] = h->host_stats[(size_t)h->p * h->q + j];
  if (n_obs_by_q)
    for (int j = 0; j < h->q; ++j) n_obs_by_q[j] = h->n_obs_q[j];
  return ST_OK;
}
extern "C" int st_xtx(st_handle h, double *xtx) {
  if (!h || !xtx) return ST_ERR_USAGE;
  std::memcpy(xtx, h->xtx.data(), h->xtx.size() * sizeof(double));
  return ST_OK;
}

extern "C" int st_yhat(st_handle h, const double *noise, uint64_t seed, uint32_t iter, double *yhat) {
  if (!h || !yhat) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  int rc = gen_or_upload_z(h, noise, seed, iter, 5u, h->d_tmp_n.p);
  if (rc) return rc;
  const int grid = (int)((h->n_all + NT - 1) / NT);
  hipLaunchKernelGGL(k_yhat, dim3(grid), dim3(NT), 0, h->stream, h->d_xb.p, h->d_w.p, h->d_tmp_n.p, h->d_mv.p, h->n_all, h->d_tsq.p,
                     h->d_tmp_n.p);
  HCHK(h, hipGetLastError());
  return download_rows(h, h->d_tmp_n.p, yhat);
}

// ---- inspection ---------------------------------------------------------------------------------------------
extern "C" int st_block_dims(st_handle h, int64_t u, int64_t *m, int64_t *P, int32_t *is_ref, int32_t *n_obs) {
  if (!h || u < 0 || u >= h->n_blocks) return ST_ERR_USAGE;
  const Blk &B = h->blks[h->blk_model2dev[u]];
  if (m) *m = B.m;
  if (P) *P = B.P;
  if (is_ref) *is_ref = B.isref;
  if (n_obs) *n_obs = B.nobs;
  return ST_OK;
}
extern "C" int st_get_block(st_handle h, int slot, int64_t u, double *negRiH, double *Ri) {
  if (!h || u < 0 || u >= h->n_blocks || slot < 0 || slot > 1) return ST_ERR_USAGE;
  const Blk &B = h->blks[h->blk_model2dev[u]];
  if (B.panel_off < 0) { h->err = "block has no observations, hence no cache"; return ST_ERR_USAGE; }
  HCHK(h, hipSetDevice(h->device));
  { const int rc0 = settle_top(h); if (rc0) return rc0; }
  std::vector<double> pan((size_t)B.m * B.ld);
  HCHK(h, hipMemcpyAsync(pan.data(), h->d_panels[h->slot_map[slot]].p + B.panel_off, pan.size() * sizeof(double), hipMemcpyDeviceToHost,
                         h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  if (negRiH)
    for (int i = 0; i < B.m; ++i)
      for (int k = 0; k < B.P; ++k) negRiH[(size_t)k * B.m + i] = pan[(size_t)i * B.ld + k];
  if (Ri) {
    if (B.isref) {
      for (int i = 0; i < B.m; ++i)
        for (int j = 0; j < B.m; ++j) Ri[(size_t)j * B.m + i] = pan[(size_t)i * B.ld + B.P + j];
    } else {
      for (int i = 0; i < B.m; ++i) Ri[i] = pan[(size_t)i * B.ld + B.P];
    }
  }
  return ST_OK;
}
extern "C" int st_get_comps(st_handle h, int slot, double *logdet_c, double *loglik_c) {
  if (!h || slot < 0 || slot > 1) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  { const int rc0 = settle_top(h); if (rc0) return rc0; }
  const int phys = h->slot_map[slot];
  std::vector<double> a(h->n_blocks), b(h->n_blocks);
  HCHK(h, hipMemcpyAsync(a.data(), h->d_logdet[phys].p, h->n_blocks * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipMemcpyAsync(b.data(), h->d_loglik[phys].p, h->n_blocks * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  for (long long i = 0; i < h->n_blocks; ++i) {
    const int u = h->blks[i].model_id;
    if (logdet_c) logdet_c[u] = a[i];
    if (loglik_c) loglik_c[u] = b[i];
  }
  return ST_OK;
}

// ---- measurement --------------------------------------------------------------------------------------------
extern "C" int st_algorithmic_bytes(st_handle h, double *out5, double *flops3) {
  if (!h || !out5) return ST_ERR_USAGE;
  double a = 0, b = 0, c = 0, msg = 0, fa = 0, fb = 0, fc = 0;
  for (auto &L : h->levels) { a += L.alg_bytes_A; b += L.alg_bytes_B; c += L.alg_bytes_C; msg += L.alg_bytes_msg; fa += L.flops_A; fb += L.flops_B; fc += L.flops_C; }
  out5[0] = a; out5[1] = b; out5[2] = c; out5[3] = msg; out5[4] = 40.0 * (double)h->n_obs;
  if (flops3) { flops3[0] = fa; flops3[1] = fb; flops3[2] = fc; }
  return ST_OK;
}
extern "C" int st_profile_enable(st_handle h, int enable) {
  if (!h) return ST_ERR_USAGE;
  h->prof = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
  return ST_OK;
}
extern "C" int st_profile_get(st_handle h, double *ms_total, int64_t *launches) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  prof_harvest(h);
  for (int f = 0; f < ST_N_KERNEL_FAMILIES; ++f) {
    if (ms_total) ms_total[f] = h->prof_ms[f];
    if (launches) launches[f] = h->prof_n[f];
    h->prof_ms[f] = 0; h->prof_n[f] = 0;
  }
  return ST_OK;
}
extern "C" int st_profile_levels(st_handle h, int32_t *n_levels, double *ms_by_level, double *bytes_by_level, int32_t cap) {
  if (!h || !n_levels) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  prof_harvest(h);
  *n_levels = h->n_actual_groups;
  // entries [0, n_levels): phase A; when cap >= 2 n_levels, entries [n_levels, 2 n_levels): phase B (k_sample*)
  const int ng = h->n_actual_groups;
  for (int g = 0; g < 2 * ng && g < cap; ++g) {
    if (ms_by_level) ms_by_level[g] = h->prof_level_n[g] ? h->prof_level_ms[g] / (double)h->prof_level_n[g] : 0.0;  // mean per launch
    if (bytes_by_level) bytes_by_level[g] = g < ng ? h->levels[g].alg_bytes_A : h->levels[g - ng].alg_bytes_B + h->levels[g - ng].alg_bytes_msg;
    h->prof_level_ms[g] = 0.0; h->prof_level_n[g] = 0;
  }
  return ST_OK;
}

// which phase-A kernel each observed level takes and the sizes that decide it (tests prove the branch they mean to reach)
extern "C" int st_level_info(st_handle h, int32_t *n_levels, int32_t *kernel, int32_t *max_m, int32_t *max_P, int32_t *n_blocks, int32_t cap) {
  if (!h || !n_levels) return ST_ERR_USAGE;
  *n_levels = h->n_actual_groups;
  for (int g = 0; g < h->n_actual_groups && g < cap; ++g) {
    const LevelInfo &L = h->levels[g];
    int k = L.big_factor ? ST_KERNEL_GENERIC_SCRATCH : ST_KERNEL_GENERIC_LDS;
    if (L.fast && h->factor_gen == 3 && L.q_nkx > 0) k = ST_KERNEL_QUAD;
    else if (L.fast) k = ST_KERNEL_MFMA;
    else if (L.bigmfma && h->factor_gen == 3) k = L.lchain ? (L.lchain_ref ? ST_KERNEL_LCHAIN_REF : ST_KERNEL_LCHAIN) : (L.wide_count > 0 ? ST_KERNEL_WIDE : ST_KERNEL_BIGMFMA);
    if (kernel) kernel[g] = k;
    if (max_m) max_m[g] = L.maxM;
    if (max_P) max_P[g] = L.maxP;
    if (n_blocks) n_blocks[g] = L.count;
  }
  return ST_OK;
}

// ---- CrossCovarianceAG10 (covariance_functions.cpp:301-355): dense n1 x n2 cross-covariance, column-major output.
// mv1 / mv2 are 1-based (as in R).  Like the reference it refuses a 1 x 1 Dmat ("Invalid Dmat for multivariate data").
extern "C" int st_cross_covariance_ag10(const double *coords1, const int64_t *mv1, int64_t n1, const double *coords2, const int64_t *mv2,
                                        int64_t n2, const double *ai1, const double *ai2, const double *phi_i, const double *thetamv,
                                        const double *Dmat, int32_t q, int32_t device, double *out) {
  if (!coords1 || !coords2 || !mv1 || !mv2 || !out || !Dmat || q < 2 || q > QMAX) {
    g_create_error = q < 2 ? "Invalid Dmat for multivariate data" : "st_cross_covariance_ag10: bad argument";
    return ST_ERR_USAGE;
  }
  if (hipSetDevice(device) != hipSuccess) { g_create_error = "no usable HIP device"; return ST_ERR_HIP; }
  CovPar cp;
  std::memset(&cp, 0, sizeof(cp));
  cp.q = q; cp.ncb = q > 2 ? 3 : 1;
  for (int j = 0; j < q; ++j) { cp.ai1[j] = ai1[j]; cp.ai2[j] = ai2[j]; cp.phi[j] = phi_i[j]; }
  for (int j = 0; j < cp.ncb; ++j) cp.tmv[j] = thetamv[j];
  for (int i = 0; i < q * q; ++i) cp.D[i] = Dmat[i];   // symmetric: layout irrelevant
  finish_covpar(&cp);
  std::vector<int> m1(n1), m2(n2);
  for (int64_t i = 0; i < n1; ++i) { m1[i] = (int)mv1[i] - 1; if (m1[i] < 0 || m1[i] >= q) { g_create_error = "mv1 out of range"; return ST_ERR_USAGE; } }
  for (int64_t i = 0; i < n2; ++i) { m2[i] = (int)mv2[i] - 1; if (m2[i] < 0 || m2[i] >= q) { g_create_error = "mv2 out of range"; return ST_ERR_USAGE; } }
  DevBuf<double> d1, d2, dout;
  DevBuf<int> dm1, dm2;
  int rc = ST_OK;
  auto bad = [&](hipError_t e) { if (e != hipSuccess) { g_create_error = hipGetErrorString(e); rc = ST_ERR_HIP; } return e != hipSuccess; };
  if (!bad(d1.alloc(2 * n1)) && !bad(d2.alloc(2 * n2)) && !bad(dout.alloc((size_t)n1 * n2)) && !bad(dm1.upload(m1)) && !bad(dm2.upload(m2)) &&
      !bad(hipMemcpy(d1.p, coords1, 2 * n1 * sizeof(double), hipMemcpyHostToDevice)) &&
      !bad(hipMemcpy(d2.p, coords2, 2 * n2 * sizeof(double), hipMemcpyHostToDevice))) {
    hipLaunchKernelGGL(k_cross_cov, dim3((unsigned)((n1 + NT - 1) / NT), (unsigned)n2), dim3(NT), 0, 0, d1.p, dm1.p, (long long)n1, d2.p, dm2.p,
                       (long long)n2, cp, dout.p);
    if (!bad(hipGetLastError())) bad(hipMemcpy(out, dout.p, (size_t)n1 * n2 * sizeof(double), hipMemcpyDeviceToHost));
  }
  d1.free(); d2.free(); dout.free(); dm1.free(); dm2.free();
  return rc;
}

// ---- running posterior means on device: call st_summary_accumulate on every saved iteration
extern "C" int st_summary_reset(st_handle h) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  if (!h->d_sum_w.p) { HCHK(h, h->d_sum_w.alloc(h->n_all)); HCHK(h, h->d_sum_yhat.alloc(h->n_all)); }
  HCHK(h, hipMemsetAsync(h->d_sum_w.p, 0, h->n_all * sizeof(double), h->stream));
  HCHK(h, hipMemsetAsync(h->d_sum_yhat.p, 0, h->n_all * sizeof(double), h->stream));
  h->n_summary = 0; h->n_draws = 0;
  return ST_OK;
}
extern "C" int st_summary_accumulate(st_handle h, uint64_t seed, uint32_t iter) {   // yhat noise: device stream 5 (spamtree_fit.cpp:384)
  if (!h) return ST_ERR_USAGE;
  if (!h->d_sum_w.p) { const int rc0 = st_summary_reset(h); if (rc0) return rc0; }
  HCHK(h, hipSetDevice(h->device));
  int rc = gen_or_upload_z(h, nullptr, seed, iter, 5u, h->d_tmp_n.p);
  if (rc) return rc;
  const int grid = (int)((h->n_all + NT - 1) / NT);
  hipLaunchKernelGGL(k_yhat, dim3(grid), dim3(NT), 0, h->stream, h->d_xb.p, h->d_w.p, h->d_tmp_n.p, h->d_mv.p, h->n_all, h->d_tsq.p, h->d_tmp_n.p);
  hipLaunchKernelGGL(k_axpy_sum, dim3(grid), dim3(NT), 0, h->stream, h->d_sum_yhat.p, h->d_tmp_n.p, h->n_all);
  hipLaunchKernelGGL(k_axpy_sum, dim3(grid), dim3(NT), 0, h->stream, h->d_sum_w.p, h->d_w.p, h->n_all);
  HCHK(h, hipGetLastError());
  if (h->n_draws < h->draws_cap) {   // keep the draw itself for the quantiles (device order, one contiguous row per draw)
    HCHK(h, hipMemcpyAsync(h->d_draws_w.p + (size_t)h->n_draws * h->n_all, h->d_w.p, h->n_all * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HCHK(h, hipMemcpyAsync(h->d_draws_yhat.p + (size_t)h->n_draws * h->n_all, h->d_tmp_n.p, h->n_all * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    h->n_draws += 1;
  }
  h->n_summary += 1;
  return ST_OK;
}
extern "C" int st_summary_reserve(st_handle h, int64_t keep) {
  if (!h || keep < 0) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  HCHK(h, hipStreamSynchronize(h->stream));
  h->d_draws_w.free(); h->d_draws_yhat.free();
  h->draws_cap = 0; h->n_draws = 0;
  if (keep == 0) return ST_OK;
  if (keep > 16384) { h->err = "st_summary_reserve: at most 16384 saved draws (one row's draws are sorted in one workgroup's LDS)"; return ST_ERR_UNSUPPORTED; }
  HCHK(h, h->d_draws_w.alloc((size_t)keep * h->n_all));
  HCHK(h, h->d_draws_yhat.alloc((size_t)keep * h->n_all));
  h->draws_cap = keep;
  return ST_OK;
}
extern "C" int st_summary_quantile(st_handle h, double q, double *w_q, double *yhat_q) {
  if (!h || !(q >= 0.0 && q <= 1.0)) return ST_ERR_USAGE;
  if (h->n_draws == 0) { h->err = "no draw stored: call st_summary_reserve before the saved iterations"; return ST_ERR_USAGE; }
  HCHK(h, hipSetDevice(h->device));
  int Kpad = 2;
  while (Kpad < h->n_draws) Kpad <<= 1;
  const int R = std::max(1, std::min(8, (int)(128 * 1024 / ((size_t)Kpad * 8))));
  const size_t lds = (size_t)R * Kpad * sizeof(double);
  (void)hipFuncSetAttribute((const void *)k_qtile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_limit);
  for (int which = 0; which < 2; ++which) {
    double *dst = which == 0 ? w_q : yhat_q;
    if (!dst) continue;
    QtArgs A;
    A.draws = which == 0 ? h->d_draws_w.p : h->d_draws_yhat.p; A.n = h->n_all; A.keep = (int)h->n_draws; A.Kpad = Kpad; A.R = R; A.q = q;
    A.out = h->d_tmp_n.p;
    hipLaunchKernelGGL(k_qtile, dim3((unsigned)((h->n_all + R - 1) / R)), dim3(NT), lds, h->stream, A);
    HCHK(h, hipGetLastError());
    const int rc = download_rows(h, h->d_tmp_n.p, dst);
    if (rc) return rc;
  }
  return ST_OK;
}
extern "C" int st_summary_get(st_handle h, double *w_mean, double *yhat_mean, int64_t *n_accumulated) {
  if (!h) return ST_ERR_USAGE;
  if (n_accumulated) *n_accumulated = h->n_summary;
  if (h->n_summary == 0 || !h->d_sum_w.p) { h->err = "no iteration accumulated"; return ST_ERR_USAGE; }
  HCHK(h, hipSetDevice(h->device));
  const double inv = 1.0 / (double)h->n_summary;
  if (w_mean) { int rc = download_rows(h, h->d_sum_w.p, w_mean); if (rc) return rc; for (long long i = 0; i < h->n_all; ++i) w_mean[i] *= inv; }
  if (yhat_mean) { int rc = download_rows(h, h->d_sum_yhat.p, yhat_mean); if (rc) return rc; for (long long i = 0; i < h->n_all; ++i) yhat_mean[i] *= inv; }
  return ST_OK;
}

