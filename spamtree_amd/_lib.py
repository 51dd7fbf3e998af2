"""ctypes binding of include/spamtree_hip.h.  No fallback: a missing library or GPU is an error."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SPAMTREE_LIB: an alternative build of the SAME library (A/B timing of kernel variants under profiles/micro); never a fallback
LIB_PATH = os.environ.get("SPAMTREE_LIB") or os.path.join(HERE, "libspamtree_hip.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int64)


class StProblem(C.Structure):
    _fields_ = [("n_all", C.c_int64), ("d", C.c_int32), ("q", C.c_int32), ("p", C.c_int32), ("n_groups", C.c_int32),
                ("n_blocks", C.c_int64), ("y", c_dp), ("X", c_dp), ("coords", c_dp), ("mv_id", c_ip),
                ("res_is_ref", c_ip), ("block_names", c_ip), ("block_groups", c_ip), ("indexing_ptr", c_ip),
                ("indexing_idx", c_ip), ("parents_ptr", c_ip), ("parents_idx", c_ip), ("children_ptr", c_ip),
                ("children_idx", c_ip)]


class StOptions(C.Structure):
    _fields_ = [("device", C.c_int32), ("reference_quirks", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32),
                ("force_generic", C.c_int32), ("reserved", C.c_int32)]


# every symbol include/spamtree_hip.h declares: name -> (restype, argtypes)
H = C.c_void_p
SIGNATURES = {
    "st_create": (C.c_int, [C.POINTER(StProblem), C.POINTER(StOptions), C.POINTER(H)]),
    "st_destroy": (C.c_int, [H]),
    "st_last_error": (C.c_char_p, [H]),
    "st_set_w": (C.c_int, [H, c_dp]),
    "st_get_w": (C.c_int, [H, c_dp]),
    "st_set_beta": (C.c_int, [H, c_dp]),
    "st_set_tausq_inv": (C.c_int, [H, c_dp]),
    "st_get_xb": (C.c_int, [H, c_dp]),
    "st_factor": (C.c_int, [H, C.c_int, c_dp, C.c_int, c_dp]),
    "st_factor_begin": (C.c_int, [H, C.c_int, c_dp, C.c_int]),
    "st_factor_ahead_levels": (C.c_int, [H]),
    "st_swap": (C.c_int, [H]),
    "st_sample_w": (C.c_int, [H, c_dp, C.c_uint64, C.c_uint32]),
    "st_loglik_w": (C.c_int, [H, C.c_int, c_dp]),
    "st_sample_w_loglik": (C.c_int, [H, c_dp, C.c_uint64, C.c_uint32, C.c_int, c_dp]),
    "st_sample_w_loglik_begin": (C.c_int, [H, c_dp, C.c_uint64, C.c_uint32, C.c_int]),
    "st_sample_w_loglik_end": (C.c_int, [H, c_dp]),
    "st_predict": (C.c_int, [H, C.c_int]),
    "st_beta_stats": (C.c_int, [H, c_dp]),
    "st_tausq_stats": (C.c_int, [H, c_dp, c_ip]),
    "st_xtx": (C.c_int, [H, c_dp]),
    "st_yhat": (C.c_int, [H, c_dp, C.c_uint64, C.c_uint32, c_dp]),
    "st_block_dims": (C.c_int, [H, C.c_int64, c_ip, c_ip, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "st_get_block": (C.c_int, [H, C.c_int, C.c_int64, c_dp, c_dp]),
    "st_get_comps": (C.c_int, [H, C.c_int, c_dp, c_dp]),
    "st_algorithmic_bytes": (C.c_int, [H, c_dp, c_dp]),
    "st_factor_ahead_enable": (C.c_int, [H, C.c_int]),
    "st_probe_peaks": (C.c_int, [C.c_int, C.c_int64, C.c_int, c_dp]),
    "st_profile_enable": (C.c_int, [H, C.c_int]),
    "st_profile_get": (C.c_int, [H, c_dp, c_ip]),
    "st_profile_levels": (C.c_int, [H, C.POINTER(C.c_int32), c_dp, c_dp, C.c_int32]),
    "st_level_info": (C.c_int, [H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int32), C.c_int32]),
    "st_synchronize": (C.c_int, [H]),
    "st_stream": (C.c_void_p, [H]),
    "st_set_stream": (C.c_int, [H, C.c_void_p]),
    "st_shard_plan": (C.c_int, [C.POINTER(StProblem), C.c_int32, c_ip, C.POINTER(C.c_int32)]),
    "st_shard_plan_opt": (C.c_int, [C.POINTER(StProblem), C.POINTER(StOptions), C.c_int32, c_ip, C.POINTER(C.c_int32)]),
    "st_shard_info": (C.c_int, [H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), c_ip, c_ip]),
    "st_factor_local": (C.c_int, [H, C.c_int, c_dp, C.c_int]),
    "st_factor_enqueue": (C.c_int, [H, C.c_int, c_dp, C.c_int]),
    "st_factor_finish": (C.c_int, [H, c_dp]),
    "st_factor_is_async": (C.c_int, [H]),
    "st_loglik_local": (C.c_int, [H, C.c_int]),
    "st_mg_pack_comps": (C.c_int, [H, C.c_int, C.POINTER(C.c_void_p), c_ip]),
    "st_mg_finish": (C.c_int, [H, c_dp]),
    "st_sample_w_local": (C.c_int, [H, c_dp, C.c_uint64, C.c_uint32]),
    "st_mg_top_region": (C.c_int, [H, C.POINTER(C.c_void_p), c_ip]),
    "st_sample_w_top": (C.c_int, [H]),
    "st_mg_pack_w": (C.c_int, [H, C.POINTER(C.c_void_p), c_ip]),
    "st_mg_unpack_w": (C.c_int, [H]),
    "st_mg_gather_w_pack": (C.c_int, [H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_ip]),
    "st_mg_gather_w_unpack": (C.c_int, [H]),
    "st_cross_covariance_ag10": (C.c_int, [c_dp, c_ip, C.c_int64, c_dp, c_ip, C.c_int64, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int32,
                                           C.c_int32, c_dp]),
    "st_summary_reset": (C.c_int, [H]),
    "st_summary_accumulate": (C.c_int, [H, C.c_uint64, C.c_uint32]),
    "st_summary_get": (C.c_int, [H, c_dp, c_dp, c_ip]),
    "st_summary_reserve": (C.c_int, [H, C.c_int64]),
    "st_summary_quantile": (C.c_int, [H, C.c_double, c_dp, c_dp]),
    "st_comm_unique_id": (C.c_int, [C.c_void_p, C.c_int32]),
    "st_comm_init": (C.c_int, [H, C.c_void_p]),
}


class StmFlags(C.Structure):
    _fields_ = [("adapting", C.c_int32), ("sample_beta", C.c_int32), ("sample_tausq", C.c_int32),
                ("sample_theta", C.c_int32), ("sample_w", C.c_int32), ("sample_predicts", C.c_int32)]


# include/spamtree_fit.h (C++ host driver)
SIGNATURES.update({
    "stm_create": (C.c_int, [C.POINTER(StProblem), C.POINTER(StOptions), c_dp, c_dp, c_dp, C.c_int, c_dp, C.c_double,
                             C.c_uint64, C.POINTER(StmFlags), C.POINTER(H)]),
    "stm_init": (C.c_int, [H]),
    "stm_destroy": (C.c_int, [H]),
    "stm_last_error": (C.c_char_p, [H]),
    "stm_handle": (C.c_void_p, [H]),
    "stm_step": (C.c_int, [H, C.c_int]),
    "stm_state": (C.c_int, [H, c_dp, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp]),
    "spamtree_mv_mcmc_c": (C.c_int, [C.POINTER(StProblem), C.POINTER(StOptions), c_dp, c_dp, C.c_int, c_dp, C.c_double,
                                     c_dp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(StmFlags), c_dp, c_dp,
                                     c_dp, c_dp, c_dp, c_dp, c_dp]),
})

# include/spamtree_tree.h (device parts of the tree builder)
c_i32p = C.POINTER(C.c_int32)
SIGNATURES.update({
    "st_tb_sort": (C.c_int, [c_dp, C.c_int64, C.c_int32, c_dp]),
    "st_tb_cell_argmin": (C.c_int, [c_ip, c_dp, c_ip, C.c_int64, C.c_int64, C.c_int32, c_ip]),
    "st_tb_nearest": (C.c_int, [c_dp, c_dp, c_i32p, C.c_int64, c_dp, c_dp, c_i32p, C.c_int64, C.c_int32, C.c_int32, c_ip]),
})

_lib = None


def load():
    """Load libspamtree_hip.so (built by spamtree_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m spamtree_amd.build` (hipcc, gfx950). "
                           "There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
