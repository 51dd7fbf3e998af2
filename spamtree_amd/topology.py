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
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

__all__ = [
    "kthresholds", "part_axis_parallel", "make_tree", "make_edges", "make_edges_limited", "prepare", "Topology", "TreeResult",
    "grid_coords",
]


def kthresholds(x: np.ndarray, k: int) -> np.ndarray:
    """k-1 order-statistic thresholds: element ``i*n//k`` of sorted ``x`` (tree_dep.cpp:16-27)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    if k <= 1 or n == 0:
        return np.zeros(0)
    xs = np.sort(x, kind="stable")
    q = (np.arange(1, k, dtype=np.int64) * n) // k
    return xs[q]


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
              cherrypick_group_locations: bool = True) -> TreeResult:
    """Recursive axis-parallel partition with one knot per fine cell (make_tree.R:1-420).

    ``coords`` n x d, ``observed`` boolean (False = NA outcome, goes to the prediction level),
    ``mv_id`` 1-based outcome id per row.  Rows are identified by their position 0..n-1 (``ix``).
    """
    from scipy.spatial import cKDTree

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
    _, gix_all = np.unique(coords, axis=0, return_inverse=True)
    gix_all = gix_all.reshape(-1)

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
        thr_knots = [kthresholds(c_av[:, i], axis_cell_size[i] * K[i] ** (res - 1)) for i in range(dd)]
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
            order = np.lexsort((ix_av[rem_idx], d2, code))
            first = np.ones(order.size, dtype=bool)
            first[1:] = code[order][1:] != code[order][:-1]
            chosen = rem_idx[order[first]]
            if cherrypick_group_locations:
                sel_g = np.zeros(gix_all.max() + 1, dtype=bool)
                sel_g[gix_all[ix_av[chosen]]] = True
                chosen = rem_idx[sel_g[gix_all[ix_av[rem_idx]]]]
            knots = np.sort(chosen)
        else:
            knots = rem_idx
        thr_res = [kthresholds(c_av[:, i], K[i] ** (res - 1)) for i in range(dd)]
        thresholds_list.append(thr_res)
        sizes_r = [t.size + 1 for t in thr_res]
        code_k = _cell_code(part_axis_parallel(c_av[knots], thr_res), sizes_r)
        present = np.unique(code_k)
        blk = max_block_number + 1 + np.searchsorted(present, code_k)
        level_block_of_code.append({int(c): int(max_block_number + 1 + j) for j, c in enumerate(present)})
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
        lut = level_block_of_code[l]
        keys = np.fromiter(lut.keys(), dtype=np.int64, count=len(lut))
        vals = np.fromiter(lut.values(), dtype=np.int64, count=len(lut))
        o = np.argsort(keys)
        keys, vals = keys[o], vals[o]
        p = np.searchsorted(keys, codes)
        p = np.clip(p, 0, keys.size - 1)
        paths[:, l] = np.where(keys[p] == codes, vals[p], 0)
    parchi = np.unique(paths, axis=0) if paths.size else np.zeros((0, n_lev), dtype=np.int64)

    all_ix = [r_ix]; all_block = [r_block]; all_res = [r_res]
    res_is_ref_l = list(res_is_ref)

    def _nearest_block(target_ix, target_block, query_ix):
        """Nearest (same-margin if requested) placed row decides the block (make_tree.R:236, 256, 345, 367)."""
        out = np.zeros(query_ix.size, dtype=np.int64)
        if cherrypick_same_margin:
            for vv in np.unique(mv_id[query_ix]):
                qsel = mv_id[query_ix] == vv
                tsel = mv_id[target_ix] == vv
                if not tsel.any():
                    tsel = np.ones(target_ix.size, dtype=bool)
                tree = cKDTree(coords[target_ix[tsel]])
                _, nn = tree.query(coords[query_ix[qsel]], k=1)
                out[qsel] = target_block[tsel][nn]
        else:
            tree = cKDTree(coords[target_ix])
            _, nn = tree.query(coords[query_ix], k=1)
            out = target_block[nn]
        return out

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
        col = np.zeros(parchi.shape[0], dtype=np.int64)
        first_new = max_block_number - uniq.size + 1
        for j, parent in enumerate(uniq):
            col[parchi[:, -1] == parent] = first_new + j
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
        col = np.zeros(parchi.shape[0], dtype=np.int64)
        for j, parent in enumerate(uniq):
            col[parchi[:, lev_col] == parent] = base + 1 + j
        parchi = np.column_stack([parchi, col])
        res_is_ref_l.append(0)

    res_is_ref = np.asarray(res_is_ref_l, dtype=np.int64)
    parchi = np.unique(parchi, axis=0)
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
                par_pairs.append(np.unique(np.column_stack([col[sel], parchimat[sel, c]]), axis=0))
        if res_is_ref[lev] == 1 and lev < L - 1:
            for c in range(lev + 1, L):
                sel = ok & (parchimat[:, c] > 0)
                pr = np.unique(np.column_stack([col[sel], parchimat[sel, c]]), axis=0)
                pr = pr[non_empty[pr[:, 1]]]
                chi_pairs.append(pr)

    def _to_lists(pairs):
        ptr = np.zeros(n_blocks + 1, dtype=np.int64)
        if not pairs:
            return ptr, np.zeros(0, dtype=np.int64)
        allp = np.unique(np.concatenate(pairs, axis=0), axis=0)    # sorted by (u, other): ascending lists
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
            par_pairs.append(np.unique(np.column_stack([col[sel], parchimat[sel, c]]), axis=0))
        if res_is_ref[lev] == 1 and lev < L - 1:
            sel = ok & (parchimat[:, lev + 1] > 0)
            pr = np.unique(np.column_stack([col[sel], parchimat[sel, lev + 1]]), axis=0)
            chi_pairs.append(pr[non_empty[pr[:, 1]]])

    def _to_lists(pairs):
        ptr = np.zeros(n_blocks + 1, dtype=np.int64)
        if not pairs:
            return ptr, np.zeros(0, dtype=np.int64)
        allp = np.unique(np.concatenate(pairs, axis=0), axis=0)
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
            limited_tree: bool = False) -> Topology:
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
                     last_not_reference, cherrypick_same_margin, cherrypick_group_locations)
    blocking = np.zeros(n, dtype=np.int64)
    res_row = np.zeros(n, dtype=np.int64)
    blocking[tree.ix] = tree.block
    res_row[tree.ix] = tree.res
    if (blocking == 0).any():
        raise ValueError("make_tree left rows without a block")
    n_blocks = int(blocking.max())

    # gix_block: index of the coordinate group inside its block (spamtree_fit.R:271-279)
    _, gix = np.unique(cs, axis=0, return_inverse=True)
    gix = gix.reshape(-1)
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
