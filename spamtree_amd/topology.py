"""Treed-DAG construction: the host-side producer of the topology contract the hot path consumes.

Mirrors (not copies) the reference's R-level preparation:
  * ``kthresholds``            <- /root/reference/src/tree_dep.cpp:16-27
  * ``part_axis_parallel``     <- /root/reference/src/tree_dep.cpp:42-67 (column_threshold + part_axis_parallel_lmt)
  * ``make_tree``              <- /root/reference/R/make_tree.R:1-420 (size law :62-165, leftovers :213-305,
                                  missing rows :317-413)
  * ``make_edges``             <- /root/reference/src/tree_dep.cpp:75-130
  * ``prepare``                <- /root/reference/R/spamtree_fit.R:196-324 (sorting, blocking, indexing,
                                  block_names / block_groups, non_empty_blocks)

Deliberate difference ("Tree builder" row): the reference picks ONE knot per fine cell
with R's ``sample()`` (make_tree.R:92), which depends on R's RNG and on dplyr row order.  Here the knot of a
fine cell is the remaining row nearest to the cell centre (ties -> lowest original row id), so the same input
always yields the same tree on any machine.  Block numbering inside a level follows the (first axis fastest)
cell order rather than R's string-sorted factor levels.  The *contract* handed to the model (0-based
``indexing``, ascending ``parents`` / ``children``, ``block_groups``, ``res_is_ref``) is the reference's.

``mvbias`` (make_tree.R:7-14): the reference draws the knot of a fine cell with probability proportional to
n_margin^(-mvbias) (sparser outcomes are preferred for the upper levels).  The deterministic counterpart here: the knot
minimises d^2 * (n_margin / n_max)^mvbias, d = distance to the cell centre -- mvbias = 0 is the plain nearest-to-centre
rule, a large mvbias picks the sparsest margin present in the cell.

``device`` (make_tree / prepare): the row-parallel steps -- order statistics for the thresholds, one knot per fine cell,
same-margin nearest placed row -- run on the GPU through include/spamtree_tree.h (``st_tb_sort``, ``st_tb_cell_argmin``,
``st_tb_nearest``); the tree is IDENTICAL to the host path's (tests/test_gpu_tree.py).  Nearest-row ties go to the lowest
row id on both paths (FNN's tie order in the reference is unspecified).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

__all__ = [
    "kthresholds", "part_axis_parallel", "make_tree", "make_edges", "make_edges_limited", "prepare", "Topology", "TreeResult",
    "grid_coords",
]


def kthresholds(x: np.ndarray, k: int, presorted: bool = False) -> np.ndarray:
    """k-1 order-statistic thresholds: element ``i*n//k`` of sorted ``x`` (tree_dep.cpp:16-27).
    ``presorted``: ``x`` is already ascending (make_tree sorts every axis once, on the host or with ``st_tb_sort``)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    if k <= 1 or n == 0:
        return np.zeros(0)
    xs = x if presorted else np.sort(x, kind="stable")
    q = (np.arange(1, k, dtype=np.int64) * n) // k
    return xs[q]


def _device_lib(device):
    if device is None:
        return None
    from . import _lib
    return _lib.load()


def _sorted_axis(x, device):
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib = _device_lib(device)
    if lib is None or x.size == 0:
        return np.sort(x, kind="stable")
    from . import _lib
    out = np.empty_like(x)
    rc = lib.st_tb_sort(x.ctypes.data_as(_lib.c_dp), x.size, int(device), out.ctypes.data_as(_lib.c_dp))
    if rc != 0:
        raise RuntimeError(f"st_tb_sort failed ({rc}): {lib.st_last_error(None).decode()}")
    return out


def _cell_argmin(code, key, ix, ncells, device):
    """Row index (into the arrays given) of the row minimising (key, ix) in every occupied cell, cells ascending."""
    lib = _device_lib(device)
    if lib is None:
        order = np.lexsort((ix, key, code))
        first = np.ones(order.size, dtype=bool)
        first[1:] = code[order][1:] != code[order][:-1]
        return order[first]
    from . import _lib
    code = np.ascontiguousarray(code, dtype=np.int64); key = np.ascontiguousarray(key, dtype=np.float64)
    ix = np.ascontiguousarray(ix, dtype=np.int64)
    out = np.empty(int(ncells), dtype=np.int64)
    rc = lib.st_tb_cell_argmin(code.ctypes.data_as(_lib.c_ip), key.ctypes.data_as(_lib.c_dp), ix.ctypes.data_as(_lib.c_ip), code.size,
                               int(ncells), int(device), out.ctypes.data_as(_lib.c_ip))
    if rc != 0:
        raise RuntimeError(f"st_tb_cell_argmin failed ({rc}): {lib.st_last_error(None).decode()}")
    return out[out >= 0]


def _nearest_rows(tc, tmv, qc, qmv, n_margins, same_margin, device):
    """For every query the index (into the targets) of the nearest target -- of the same margin when requested and present --
    ties to the lowest target index."""
    tc = np.ascontiguousarray(tc, dtype=np.float64); qc = np.ascontiguousarray(qc, dtype=np.float64)
    if not same_margin:
        tmv = np.zeros(tc.shape[0], dtype=np.int64); qmv = np.zeros(qc.shape[0], dtype=np.int64); n_margins = 1
    lib = _device_lib(device)
    if lib is not None and tc.shape[0] > 0 and qc.shape[0] > 0:
        from . import _lib
        tx, ty = np.ascontiguousarray(tc[:, 0]), np.ascontiguousarray(tc[:, 1])
        qx, qy = np.ascontiguousarray(qc[:, 0]), np.ascontiguousarray(qc[:, 1])
        t32, q32 = np.ascontiguousarray(tmv, dtype=np.int32), np.ascontiguousarray(qmv, dtype=np.int32)
        out = np.empty(qc.shape[0], dtype=np.int64)
        rc = lib.st_tb_nearest(tx.ctypes.data_as(_lib.c_dp), ty.ctypes.data_as(_lib.c_dp), t32.ctypes.data_as(_lib.c_i32p), tc.shape[0],
                               qx.ctypes.data_as(_lib.c_dp), qy.ctypes.data_as(_lib.c_dp), q32.ctypes.data_as(_lib.c_i32p), qc.shape[0],
                               int(n_margins), int(device), out.ctypes.data_as(_lib.c_ip))
        if rc != 0:
            raise RuntimeError(f"st_tb_nearest failed ({rc}): {lib.st_last_error(None).decode()}")
        return out
    from scipy.spatial import cKDTree
    out = np.zeros(qc.shape[0], dtype=np.int64)
    for vv in np.unique(qmv):
        qsel = np.nonzero(qmv == vv)[0]
        tsel = np.nonzero(tmv == vv)[0]
        if tsel.size == 0:
            tsel = np.arange(tc.shape[0])
        tree = cKDTree(tc[tsel])
        todo = np.arange(qsel.size)
        k = int(min(9, tsel.size))
        while todo.size:
            _, nn = tree.query(qc[qsel[todo]], k=k)
            nn = nn.reshape(todo.size, k)
            # exact squared distances in the arithmetic the device uses, ties to the lowest target index
            cand = tsel[nn]
            dx = qc[qsel[todo], 0][:, None] - tc[cand, 0]
            dy = qc[qsel[todo], 1][:, None] - tc[cand, 1]
            d2 = dx * dx + dy * dy
            best = d2.min(axis=1)
            out[qsel[todo]] = np.where(d2 == best[:, None], cand, np.iinfo(np.int64).max).min(axis=1)
            # the device compares ALL targets: a query whose k candidates all tie with the best may have more ties beyond them
            # (co-located outcomes x equidistant grid neighbours: 12 ties with q = 3) -- ask for more until one is farther
            if k >= tsel.size:
                break
            todo = todo[d2.max(axis=1) == best]
            k = int(min(4 * k, tsel.size))
    return out


def part_axis_parallel(coords: np.ndarray, thresholds: Sequence[np.ndarray]) -> np.ndarray:
    """Per-axis cell number (1-based) = 1 + #{thresholds <= x}  (tree_dep.cpp:42-67)."""
    coords = np.asarray(coords, dtype=np.float64)
    out = np.empty(coords.shape, dtype=np.int64)
    for j in range(coords.shape[1]):
        thr = np.sort(np.asarray(thresholds[j], dtype=np.float64))
        out[:, j] = 1 + np.searchsorted(thr, coords[:, j], side="right")
    return out


def _cell_code(cells: np.ndarray, sizes: Sequence[int]) -> np.ndarray:
    """Combine per-axis cell numbers into one code, first axis fastest."""
    code = np.zeros(cells.shape[0], dtype=np.int64)
    mult = 1
    for j in range(cells.shape[1]):
        code += (cells[:, j] - 1) * mult
        mult *= int(sizes[j])
    return code


def _unique_rows(a: np.ndarray) -> np.ndarray:
    """Sorted unique rows of a small-integer matrix (what ``np.unique(a, axis=0)`` returns, by lexsort + adjacent compare
    instead of a structured-dtype sort; two non-negative columns go through one combined 64-bit key)."""
    a = np.asarray(a)
    if a.shape[0] == 0:
        return a.copy()
    if a.shape[1] == 2 and a.min() >= 0 and int(a[:, 0].max()) < (1 << 31) and int(a[:, 1].max()) < (1 << 31):
        key = np.unique((a[:, 0].astype(np.int64) << 32) | a[:, 1].astype(np.int64))
        return np.column_stack([key >> 32, key & 0xffffffff]).astype(a.dtype)
    order = np.lexsort(tuple(a[:, j] for j in range(a.shape[1] - 1, -1, -1)))
    s_ = a[order]
    keep = np.ones(s_.shape[0], dtype=bool)
    keep[1:] = np.any(s_[1:] != s_[:-1], axis=1)
    return s_[keep]


def _coord_groups(coords: np.ndarray) -> np.ndarray:
    """Index of every row's coordinate group (rows with identical coordinates), numbered in lexicographic coordinate order
    (what ``np.unique(coords, axis=0, return_inverse=True)`` returns, without its structured-dtype sort)."""
    n, dd = coords.shape
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    order = np.lexsort(tuple(coords[:, j] for j in range(dd - 1, -1, -1)))
    cs = coords[order]
    new = np.ones(n, dtype=bool)
    new[1:] = np.any(cs[1:] != cs[:-1], axis=1)
    g = np.empty(n, dtype=np.int64)
    g[order] = np.cumsum(new) - 1
    return g


@dataclass
class TreeResult:
    """Output of :func:`make_tree` (counterpart of make_tree.R:416-419)."""
    ix: np.ndarray            # original row id of every row that was placed
    block: np.ndarray         # 1-based block id of that row
    res: np.ndarray           # level ("res") of that row, start_level+1 ...
    parchi_map: np.ndarray    # unique root->leaf paths, one column per level, 1-based block ids, 0 = NA
    res_is_ref: np.ndarray    # one flag per level column
    thresholds: list = field(default_factory=list)


def make_tree(coords: np.ndarray, observed: np.ndarray, mv_id: np.ndarray,
              axis_cell_size: Sequence[int] = (5, 5), K: Sequence[int] = (2, 2),
              start_level: int = 0, tree_depth: float = np.inf,
              last_not_reference: bool = True,
              cherrypick_same_margin: bool = True,
              cherrypick_group_locations: bool = True, mvbias: float = 0.0, device: Optional[int] = None) -> TreeResult:
    """Recursive axis-parallel partition with one knot per fine cell (make_tree.R:1-420).

    ``coords`` n x d, ``observed`` boolean (False = NA outcome, goes to the prediction level),
    ``mv_id`` 1-based outcome id per row.  Rows are identified by their position 0..n-1 (``ix``).
    """
    coords = np.asarray(coords, dtype=np.float64)
    observed = np.asarray(observed, dtype=bool)
    mv_id = np.asarray(mv_id, dtype=np.int64)
    n_all, dd = coords.shape
    axis_cell_size = [int(a) for a in axis_cell_size]
    K = [int(k) for k in K]
    max_res = start_level + tree_depth

    ix_av = np.nonzero(observed)[0]
    ix_mi = np.nonzero(~observed)[0]
    c_av = coords[ix_av]
    lo = c_av.min(axis=0) if ix_av.size else np.zeros(dd)
    hi = c_av.max(axis=0) if ix_av.size else np.ones(dd)

    # co-location groups (make_tree.R:95-100 joins by coordinates)
    gix_all = _coord_groups(coords)
    # every axis of the available sample sorted ONCE: all thresholds of all levels are order statistics of it
    sorted_axes = [_sorted_axis(c_av[:, i], device) for i in range(dd)]
    # mvbias (make_tree.R:12-22): weight n_margin^(-mvbias) per outcome, here as a factor on the squared distance
    n_marg = int(mv_id.max()) if mv_id.size else 1
    cnt_marg = np.bincount(mv_id[ix_av] - 1, minlength=n_marg).astype(np.float64)
    marg_factor = np.ones(n_marg) if mvbias == 0 else (np.maximum(cnt_marg, 1.0) / max(cnt_marg.max(), 1.0)) ** float(mvbias)

    remaining = np.ones(ix_av.size, dtype=bool)        # over available rows ("cx")
    ref_ix: List[np.ndarray] = []
    ref_block: List[np.ndarray] = []
    ref_res: List[np.ndarray] = []
    level_cells_of_avail: List[np.ndarray] = []        # tessellation code of every available row per level
    level_sizes: List[int] = []
    level_block_of_code: List[dict] = []
    thresholds_list = []
    max_block_number = 0
    res = start_level + 1
    res_ix = 1

    while res <= max_res and remaining.any():
        n_rem = int(remaining.sum())
        thr_knots = [kthresholds(sorted_axes[i], axis_cell_size[i] * K[i] ** (res - 1), presorted=True) for i in range(dd)]
        sizes_k = [t.size + 1 for t in thr_knots]
        grid_size = int(np.prod(sizes_k))
        rem_idx = np.nonzero(remaining)[0]
        if grid_size < n_rem:
            cells = part_axis_parallel(c_av[rem_idx], thr_knots)
            code = _cell_code(cells, sizes_k)
            # cell centres from the bounding thresholds (domain min/max at the rim)
            d2 = np.zeros(rem_idx.size)
            for i in range(dd):
                edges = np.concatenate(([lo[i]], thr_knots[i], [hi[i]]))
                centre = 0.5 * (edges[cells[:, i] - 1] + edges[cells[:, i]])
                d2 += (c_av[rem_idx, i] - centre) ** 2
            if mvbias != 0:
                d2 = d2 * marg_factor[mv_id[ix_av[rem_idx]] - 1]
            chosen = rem_idx[_cell_argmin(code, d2, ix_av[rem_idx], grid_size, device)]
            if cherrypick_group_locations:
                sel_g = np.zeros(gix_all.max() + 1, dtype=bool)
                sel_g[gix_all[ix_av[chosen]]] = True
                chosen = rem_idx[sel_g[gix_all[ix_av[rem_idx]]]]
            knots = np.sort(chosen)
        else:
            knots = rem_idx
        thr_res = [kthresholds(sorted_axes[i], K[i] ** (res - 1), presorted=True) for i in range(dd)]
        thresholds_list.append(thr_res)
        sizes_r = [t.size + 1 for t in thr_res]
        code_k = _cell_code(part_axis_parallel(c_av[knots], thr_res), sizes_r)
        present = np.unique(code_k)
        blk = max_block_number + 1 + np.searchsorted(present, code_k)
        level_block_of_code.append((present, int(max_block_number + 1)))   # block of cell code c = base + rank of c in `present`
        max_block_number = int(blk.max())
        ref_ix.append(ix_av[knots]); ref_block.append(blk); ref_res.append(np.full(knots.size, res))
        remaining[knots] = False
        level_cells_of_avail.append(_cell_code(part_axis_parallel(c_av, thr_res), sizes_r))
        level_sizes.append(int(np.prod(sizes_r)))
        res += 1
        res_ix += 1

    n_lev = res_ix - 1
    res_is_ref = np.ones(n_lev, dtype=np.int64)
    if last_not_reference and (res < max_res) and n_lev > 0:
        res_is_ref[-1] = 0

    r_ix = np.concatenate(ref_ix) if ref_ix else np.zeros(0, dtype=np.int64)
    r_block = np.concatenate(ref_block) if ref_block else np.zeros(0, dtype=np.int64)
    r_res = np.concatenate(ref_res) if ref_res else np.zeros(0, dtype=np.int64)

    # paths: for every placed row, the block it falls in at each level (0 where that cell holds no knot)
    pos_in_av = np.full(n_all, -1, dtype=np.int64)
    pos_in_av[ix_av] = np.arange(ix_av.size)
    paths = np.zeros((r_ix.size, n_lev), dtype=np.int64)
    for l in range(n_lev):
        codes = level_cells_of_avail[l][pos_in_av[r_ix]]
        keys, base = level_block_of_code[l]
        p = np.clip(np.searchsorted(keys, codes), 0, keys.size - 1)
        paths[:, l] = np.where(keys[p] == codes, base + p, 0)
    parchi = _unique_rows(paths) if paths.size else np.zeros((0, n_lev), dtype=np.int64)

    all_ix = [r_ix]; all_block = [r_block]; all_res = [r_res]
    res_is_ref_l = list(res_is_ref)

    def _nearest_block(target_ix, target_block, query_ix):
        """Nearest (same-margin if requested) placed row decides the block (make_tree.R:236, 256, 345, 367)."""
        nn = _nearest_rows(coords[target_ix], mv_id[target_ix] - 1, coords[query_ix], mv_id[query_ix] - 1, n_marg,
                           cherrypick_same_margin, device)
        return target_block[nn]

    # leftovers (only when tree_depth is finite): one extra non-reference level under the deepest level
    if remaining.any():
        left_ix = ix_av[remaining]
        top = r_res.max()
        sub = r_res == top
        pb = _nearest_block(r_ix[sub], r_block[sub], left_ix)
        uniq, inv = np.unique(pb, return_inverse=True)
        blk = max_block_number + 1 + inv
        max_block_number = int(blk.max())
        all_ix.append(left_ix); all_block.append(blk); all_res.append(np.full(left_ix.size, top + 1))
        first_new = max_block_number - uniq.size + 1
        pp_ = np.clip(np.searchsorted(uniq, parchi[:, -1]), 0, uniq.size - 1)
        col = np.where(uniq[pp_] == parchi[:, -1], first_new + pp_, 0).astype(np.int64)
        parchi = np.column_stack([parchi, col])
        res_is_ref_l.append(0)

    if len(res_is_ref_l) == 1:          # a single observed level is a reference level (make_tree.R:307-309: the rule is
        res_is_ref_l[0] = 1             # applied BEFORE the level of the missing rows is appended, :317-413)

    # missing rows: their own last level, grouped by the block of the nearest deepest-level row
    if ix_mi.size:
        cur_ix = np.concatenate(all_ix); cur_block = np.concatenate(all_block); cur_res = np.concatenate(all_res)
        top = r_res.max()
        sub_r = r_res == top
        pb = _nearest_block(r_ix[sub_r], r_block[sub_r], ix_mi)
        uniq, inv = np.unique(pb, return_inverse=True)
        base = int(cur_block.max())
        blk = base + 1 + inv
        miss_res = int(cur_res.max()) + 1
        all_ix.append(ix_mi); all_block.append(blk); all_res.append(np.full(ix_mi.size, miss_res))
        lev_col = int(top - (start_level + 1))          # column of the parent level in parchi
        pp_ = np.clip(np.searchsorted(uniq, parchi[:, lev_col]), 0, uniq.size - 1)
        col = np.where(uniq[pp_] == parchi[:, lev_col], base + 1 + pp_, 0).astype(np.int64)
        parchi = np.column_stack([parchi, col])
        res_is_ref_l.append(0)

    res_is_ref = np.asarray(res_is_ref_l, dtype=np.int64)
    parchi = _unique_rows(parchi)
    return TreeResult(ix=np.concatenate(all_ix), block=np.concatenate(all_block), res=np.concatenate(all_res),
                      parchi_map=parchi, res_is_ref=res_is_ref, thresholds=thresholds_list)


def make_edges(parchimat: np.ndarray, non_empty_blocks: np.ndarray, res_is_ref: np.ndarray):
    """parents(u) = ancestor blocks on reference levels; children(u) = all non-empty descendants.

    Same contract as /root/reference/src/tree_dep.cpp:75-130: 0-based ids, ascending; ``parchimat`` holds
    1-based block ids with 0 for NA; ``non_empty_blocks`` is 1-based.
    """
    parchimat = np.asarray(parchimat, dtype=np.int64)
    L = parchimat.shape[1]
    n_blocks = int(parchimat.max())
    res_is_ref = np.asarray(res_is_ref, dtype=np.int64)
    non_empty = np.zeros(n_blocks + 1, dtype=bool)
    non_empty[np.asarray(non_empty_blocks, dtype=np.int64)] = True
    ref_cols = np.nonzero(res_is_ref == 1)[0]

    par_pairs = []     # (u, parent)
    chi_pairs = []     # (u, child)
    for lev in range(L):
        col = parchimat[:, lev]
        ok = col > 0
        if lev > 0:
            cols = ref_cols[ref_cols < lev] if ref_cols.size > 0 else np.arange(lev)
            for c in cols:
                sel = ok & (parchimat[:, c] > 0)
                par_pairs.append(_unique_rows(np.column_stack([col[sel], parchimat[sel, c]])))
        if res_is_ref[lev] == 1 and lev < L - 1:
            for c in range(lev + 1, L):
                sel = ok & (parchimat[:, c] > 0)
                pr = _unique_rows(np.column_stack([col[sel], parchimat[sel, c]]))
                pr = pr[non_empty[pr[:, 1]]]
                chi_pairs.append(pr)

    def _to_lists(pairs):
        ptr = np.zeros(n_blocks + 1, dtype=np.int64)
        if not pairs:
            return ptr, np.zeros(0, dtype=np.int64)
        allp = _unique_rows(np.concatenate(pairs, axis=0))    # sorted by (u, other): ascending lists
        cnt = np.bincount(allp[:, 0] - 1, minlength=n_blocks)
        ptr[1:] = np.cumsum(cnt)
        return ptr, allp[:, 1] - 1

    par_ptr, par_idx = _to_lists(par_pairs)
    chi_ptr, chi_idx = _to_lists(chi_pairs)
    return (par_ptr, par_idx), (chi_ptr, chi_idx)


def make_edges_limited(parchimat: np.ndarray, non_empty_blocks: np.ndarray, res_is_ref: np.ndarray):
    """``limited_tree = TRUE``: parents(u) = the block of u's rows on the LAST reference level above u (one block),
    children(u) = the non-empty blocks of u's rows on the NEXT level only.

    Same contract as /root/reference/src/tree_dep.cpp:133-186 (selected at /root/reference/R/spamtree_fit.R:310-311):
    0-based ids, ascending; ``parchimat`` holds 1-based block ids with 0 for NA; ``non_empty_blocks`` is 1-based.
    """
    parchimat = np.asarray(parchimat, dtype=np.int64)
    L = parchimat.shape[1]
    n_blocks = int(parchimat.max())
    res_is_ref = np.asarray(res_is_ref, dtype=np.int64)
    non_empty = np.zeros(n_blocks + 1, dtype=bool)
    non_empty[np.asarray(non_empty_blocks, dtype=np.int64)] = True
    ref_cols = np.nonzero(res_is_ref == 1)[0]
    par_pairs, chi_pairs = [], []
    for lev in range(L):
        col = parchimat[:, lev]
        ok = col > 0
        if lev > 0:
            cols = ref_cols[ref_cols < lev] if ref_cols.size > 0 else np.arange(lev)
            c = int(cols[-1])
            sel = ok & (parchimat[:, c] > 0)
            par_pairs.append(_unique_rows(np.column_stack([col[sel], parchimat[sel, c]])))
        if res_is_ref[lev] == 1 and lev < L - 1:
            sel = ok & (parchimat[:, lev + 1] > 0)
            pr = _unique_rows(np.column_stack([col[sel], parchimat[sel, lev + 1]]))
            chi_pairs.append(pr[non_empty[pr[:, 1]]])

    def _to_lists(pairs):
        ptr = np.zeros(n_blocks + 1, dtype=np.int64)
        if not pairs:
            return ptr, np.zeros(0, dtype=np.int64)
        allp = _unique_rows(np.concatenate(pairs, axis=0))
        cnt = np.bincount(allp[:, 0] - 1, minlength=n_blocks)
        ptr[1:] = np.cumsum(cnt)
        return ptr, allp[:, 1] - 1

    par_ptr, par_idx = _to_lists(par_pairs)
    chi_ptr, chi_idx = _to_lists(chi_pairs)
    return (par_ptr, par_idx), (chi_ptr, chi_idx)


@dataclass
class Topology:
    """Everything `spamtree_mv_mcmc` receives from R (spamtree_fit.R:327-362), in the sorted row order.

    Lists of index vectors are CSR pairs ``(ptr, idx)``; all ids 0-based except ``block_names`` (1-based, as in R).
    """
    n: int
    q: int
    sort_ix: np.ndarray           # original row id of sorted row i
    coords: np.ndarray            # n x d (sorted order)
    mv_id: np.ndarray             # 1-based
    blocking: np.ndarray          # 1-based block id per row
    gix_block: np.ndarray
    res_is_ref: np.ndarray
    parents_ptr: np.ndarray
    parents_idx: np.ndarray
    children_ptr: np.ndarray
    children_idx: np.ndarray
    block_names: np.ndarray       # 1-based
    block_groups: np.ndarray      # level ("res") of block id-1
    indexing_ptr: np.ndarray
    indexing_idx: np.ndarray
    parchi_map: np.ndarray

    @property
    def n_blocks(self) -> int:
        return int(self.block_names.size)

    def indexing(self, u: int) -> np.ndarray:
        return self.indexing_idx[self.indexing_ptr[u]:self.indexing_ptr[u + 1]]

    def parents(self, u: int) -> np.ndarray:
        return self.parents_idx[self.parents_ptr[u]:self.parents_ptr[u + 1]]

    def children(self, u: int) -> np.ndarray:
        return self.children_idx[self.children_ptr[u]:self.children_ptr[u + 1]]


def prepare(y: np.ndarray, coords: np.ndarray, mv_id: Optional[np.ndarray] = None,
            cell_size: int = 25, K: Optional[Sequence[int]] = None, start_level: int = 0,
            tree_depth: float = np.inf, last_not_reference: bool = True,
            cherrypick_same_margin: bool = True, cherrypick_group_locations: bool = True,
            limited_tree: bool = False, mvbias: float = 0.0, device: Optional[int] = None) -> Topology:
    """Row sorting, tree, edges and indexing exactly as `spamtree()` hands them to C++ (spamtree_fit.R:196-324).

    ``y`` may contain NaN (= NA).  Returns arrays in the *sorted* row order (by coordinates, then original id).
    """
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    coords = np.asarray(coords, dtype=np.float64)
    n, dd = coords.shape
    mv_id = np.ones(n, dtype=np.int64) if mv_id is None else np.asarray(mv_id, dtype=np.int64)
    K = [2] * dd if K is None else list(K)
    axis_size = int(round(cell_size ** (1.0 / dd))) if np.isscalar(cell_size) else None
    axis_cell_size = [axis_size] * dd if axis_size is not None else list(cell_size)

    # arrange(Var1..Vard, ix)   (spamtree_fit.R:214, 267-269)
    keys = [np.arange(n)] + [coords[:, j] for j in range(dd - 1, -1, -1)]
    sort_ix = np.lexsort(keys)
    cs = coords[sort_ix]; ys = y[sort_ix]; ms = mv_id[sort_ix]
    tree = make_tree(cs, np.isfinite(ys), ms, axis_cell_size, K, start_level, tree_depth,
                     last_not_reference, cherrypick_same_margin, cherrypick_group_locations, mvbias, device)
    blocking = np.zeros(n, dtype=np.int64)
    res_row = np.zeros(n, dtype=np.int64)
    blocking[tree.ix] = tree.block
    res_row[tree.ix] = tree.res
    if (blocking == 0).any():
        raise ValueError("make_tree left rows without a block")
    n_blocks = int(blocking.max())

    # gix_block: index of the coordinate group inside its block (spamtree_fit.R:271-279)
    gix = _coord_groups(cs)
    gix_block = np.zeros(n, dtype=np.int64)
    o = np.lexsort((gix, blocking))
    bs, gs = blocking[o], gix[o]
    newb = np.ones(n, dtype=bool); newb[1:] = bs[1:] != bs[:-1]
    newg = newb.copy(); newg[1:] |= gs[1:] != gs[:-1]
    run = np.cumsum(newg)
    start_of_block = np.maximum.accumulate(np.where(newb, run, 0))
    gix_block[o] = run - start_of_block + 1

    # indexing = split(0-based row ids, block)   (spamtree_fit.R:324)
    order = np.argsort(blocking, kind="stable")
    cnt = np.bincount(blocking - 1, minlength=n_blocks)
    idx_ptr = np.zeros(n_blocks + 1, dtype=np.int64); idx_ptr[1:] = np.cumsum(cnt)
    idx = order.astype(np.int64)

    obs_cnt = np.bincount(blocking - 1, weights=np.isfinite(ys).astype(np.float64), minlength=n_blocks)
    non_empty_blocks = np.nonzero(obs_cnt > 0)[0] + 1
    edges = make_edges_limited if limited_tree else make_edges          # spamtree_fit.R:310-314
    (pp, pi), (cp, ci) = edges(tree.parchi_map, non_empty_blocks, tree.res_is_ref)

    block_groups = np.zeros(n_blocks, dtype=np.int64)
    block_groups[blocking - 1] = res_row
    _, first = np.unique(blocking, return_index=True)
    block_names = blocking[np.sort(first)]          # order of first appearance, as `unique()` gives in R
    return Topology(n=n, q=int(np.unique(ms).size), sort_ix=sort_ix, coords=cs, mv_id=ms, blocking=blocking,
                    gix_block=gix_block, res_is_ref=tree.res_is_ref, parents_ptr=pp, parents_idx=pi,
                    children_ptr=cp, children_idx=ci, block_names=block_names, block_groups=block_groups,
                    indexing_ptr=idx_ptr, indexing_idx=idx, parchi_map=tree.parchi_map)


def grid_coords(side: int, q: int = 1):
    """Regular ``side x side`` grid on [0,1]^2 replicated per outcome (SURVEY.md section 8d synthetic inputs)."""
    xs = np.linspace(0.0, 1.0, side)
    g = np.stack(np.meshgrid(xs, xs, indexing="ij"), axis=-1).reshape(-1, 2)
    coords = np.tile(g, (q, 1))
    mv_id = np.repeat(np.arange(1, q + 1), side * side)
    return coords, mv_id
