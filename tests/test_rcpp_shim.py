"""CPU: the compile-guarded R shim (spamtree_amd/csrc/rcpp_exports.cpp) parses and type-checks -- against a STUB of the Rcpp /
Armadillo names it uses (tests/stubs/RcppArmadillo.h; R, Rcpp and RcppArmadillo are not in the image).  A syntax and signature
check only: nothing is linked or run, nothing numeric is pinned.  What it does pin: the exported names, the 35 / 9 argument
lists and the ten returned list names of the reference (/root/reference/src/spamtree_fit.cpp:5-54, 403-414;
/root/reference/src/covariance_functions.cpp:301-309; /root/reference/src/RcppExports.cpp:20-38, 112-154)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "spamtree_amd", "csrc", "rcpp_exports.cpp")


def test_shim_type_checks_against_the_stub_header():
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-DSPAMTREE_WITH_RCPP", "-I", os.path.join(ROOT, "tests", "stubs"),
                        "-I", os.path.join(ROOT, "include"), SRC], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _args(src, name):
    m = re.search(name + r"\((.*?)\)\s*\{", src, flags=re.S)
    depth, parts, cur = 0, [], ""
    for ch in m.group(1):
        if ch in "<(":
            depth += 1
        if ch in ">)":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip()); cur = ""
        else:
            cur += ch
    parts.append(cur.strip())
    return [re.sub(r"\s*=.*$", "", p).split()[-1].lstrip("&*") for p in parts]


def test_exported_signatures_and_returned_names_are_the_references():
    src = open(SRC).read()
    mcmc = _args(src, r"Rcpp::List spamtree_mv_mcmc")
    assert mcmc == ["y", "X", "Z", "coords", "mv_id", "blocking", "gix_block", "res_is_ref", "parents", "children", "limited_tree",
                    "layer_names", "layer_gibbs_group", "indexing", "set_unif_bounds_in", "start_w", "theta", "beta", "tausq", "mcmcsd",
                    "mcmc_keep", "mcmc_burn", "mcmc_thin", "num_threads", "use_alg", "adapting", "main_verbose", "verbose", "debug",
                    "printall", "sample_beta", "sample_tausq", "sample_theta", "sample_w", "sample_predicts"]      # 35
    cc = _args(src, r"arma::mat CrossCovarianceAG10")
    assert cc == ["coords1", "mv1", "coords2", "mv2", "ai1", "ai2", "phi_i", "thetamv", "Dmat"]                   # 9
    names = re.findall(r'Rcpp::Named\("(\w+)"\)', src)
    assert names[-10:] == ["w_mcmc", "yhat_mcmc", "beta_mcmc", "tausq_mcmc", "theta_mcmc", "paramsd", "block_ct_obs", "indexing",
                           "parents_indexing", "mcmc_time"]
    assert src.count("// [[Rcpp::export]]") == 2
