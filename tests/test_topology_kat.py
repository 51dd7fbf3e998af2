"""CPU: known answers for the tree builder's primitives, WORKED BY HAND from the reference's source (not produced by running
this repository's code): kthresholds (/root/reference/src/tree_dep.cpp:16-27), column_threshold / part_axis_parallel_lmt
(:42-67), make_edges (:75-130), make_edges_limited (:133-186); plus the invariants of the deterministic knot rule, of the
mvbias weighting (/root/reference/R/make_tree.R:7-22) and of the nearest-row tie rule."""
import numpy as np

from spamtree_amd import topology as tp


def test_kthresholds_known_answers():
    # res(i-1) = x sorted at position floor(i * n / k): n = 10, k = 4 -> positions 2, 5, 7
    x = np.array([9.0, 1.0, 8.0, 2.0, 7.0, 3.0, 6.0, 4.0, 5.0, 0.0])
    assert tp.kthresholds(x, 4).tolist() == [2.0, 5.0, 7.0]
    # n = 7, k = 3 -> positions 2, 4 of (10, 20, ..., 70)
    assert tp.kthresholds(np.arange(70.0, 0.0, -10.0), 3).tolist() == [30.0, 50.0]
    assert tp.kthresholds(x, 1).size == 0
    assert tp.kthresholds(np.sort(x), 4, presorted=True).tolist() == [2.0, 5.0, 7.0]


def test_part_axis_parallel_known_answers():
    # cell = 1 + #{thresholds t : x >= t}
    c = np.array([[0.0, 5.0], [1.0, 5.0], [0.99, 4.99], [3.0, 9.0], [-1.0, 0.0]])
    thr = [np.array([1.0, 2.0]), np.array([5.0])]
    assert tp.part_axis_parallel(c, thr).tolist() == [[1, 2], [2, 2], [1, 1], [3, 2], [1, 1]]


PARCHI = np.array([[1, 2, 4],      # root 1; level-2 blocks 2, 3; level-3 blocks 4, 5, 6 (6 under 3), 0 = no knot on that level
                   [1, 2, 5],
                   [1, 3, 6],
                   [1, 3, 0]])


def test_make_edges_known_answers():
    # all three levels reference levels, every block non-empty
    (pp, pi), (cp, ci) = tp.make_edges(PARCHI, np.arange(1, 7), np.array([1, 1, 1]))
    par = [pi[pp[u]:pp[u + 1]].tolist() for u in range(6)]
    chi = [ci[cp[u]:cp[u + 1]].tolist() for u in range(6)]
    assert par == [[], [0], [0], [0, 1], [0, 1], [0, 2]]                   # 0-based, ascending = root first
    assert chi == [[1, 2, 3, 4, 5], [3, 4], [5], [], [], []]              # ALL descendants (:102-106), none for the last level
    # block 5 (1-based) empty (to be predicted): it keeps its parents and is nobody's child (:77, 106); last level not reference
    (pp, pi), (cp, ci) = tp.make_edges(PARCHI, np.array([1, 2, 3, 4, 6]), np.array([1, 1, 0]))
    assert [pi[pp[u]:pp[u + 1]].tolist() for u in range(6)][4] == [0, 1]
    assert [ci[cp[u]:cp[u + 1]].tolist() for u in range(6)] == [[1, 2, 3, 5], [3], [5], [], [], []]


def test_make_edges_limited_known_answers():
    (pp, pi), (cp, ci) = tp.make_edges_limited(PARCHI, np.arange(1, 7), np.array([1, 1, 1]))
    assert [pi[pp[u]:pp[u + 1]].tolist() for u in range(6)] == [[], [0], [0], [1], [1], [2]]   # the one block on the last reference level above
    assert [ci[cp[u]:cp[u + 1]].tolist() for u in range(6)] == [[1, 2], [3, 4], [5], [], [], []]   # the next level only


def test_unique_rows_and_coordinate_groups_match_numpy():
    rng = np.random.default_rng(0)
    for shape, hi in (((500, 2), 9), ((400, 5), 3), ((1, 3), 2), ((0, 4), 2)):
        a = rng.integers(0, hi, size=shape)
        assert np.array_equal(tp._unique_rows(a), np.unique(a, axis=0)) or a.shape[0] == 0
    c = rng.integers(0, 6, size=(300, 2)).astype(np.float64) / 7.0
    _, inv = np.unique(c, axis=0, return_inverse=True)
    assert np.array_equal(tp._coord_groups(c), inv.reshape(-1))


def test_knot_rule_one_knot_per_cell_nearest_to_centre():
    rng = np.random.default_rng(1)
    code = rng.integers(0, 12, size=400)
    code[code == 7] = 3                                     # an empty cell
    key = rng.uniform(size=400).round(2)                    # ties on purpose
    ix = rng.permutation(400)
    rows = tp._cell_argmin(code, key, ix, 12, None)
    assert np.array_equal(code[rows], np.setdiff1d(np.arange(12), [7]))
    for r in rows:
        sel = np.nonzero(code == code[r])[0]
        best = key[sel].min()
        assert key[r] == best and ix[r] == ix[sel][key[sel] == best].min()


def test_nearest_row_same_margin_and_tie_rule():
    # four targets at the corners of a unit square, the query in the centre: an exact four-way tie -> lowest index of the margin
    tc = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.5, 0.4]])
    tmv = np.array([1, 1, 0, 0, 2])
    qc = np.array([[0.5, 0.5], [0.5, 0.5], [0.5, 0.5], [0.9, 0.1]])
    qmv = np.array([0, 1, 3, 2])                            # margin 3 has no target: all targets compete (make_tree.R:256 fallback)
    nn = tp._nearest_rows(tc, tmv, qc, qmv, 4, True, None)
    assert nn.tolist() == [2, 0, 4, 4]
    assert tp._nearest_rows(tc, tmv, qc, qmv, 4, False, None).tolist() == [4, 4, 4, 1]


def test_mvbias_prefers_the_sparser_margin_for_the_upper_levels():
    """make_tree.R:8-14: mvbias > 0 = 'prefer picking sparser margins for lower levels of the tree' (weights n_margin^-mvbias)."""
    rng = np.random.default_rng(3)
    n1, n2 = 1500, 150                                      # outcome 2 is ten times sparser, at its own locations
    coords = np.vstack([rng.uniform(size=(n1, 2)), rng.uniform(size=(n2, 2))])
    mv = np.concatenate([np.ones(n1, dtype=np.int64), 2 * np.ones(n2, dtype=np.int64)])
    y = rng.standard_normal(n1 + n2)
    share = []
    for mvbias in (0.0, 2.0):
        t = tp.prepare(y, coords, mv, mvbias=mvbias)
        top = t.block_groups[t.blocking - 1] <= 2           # rows placed on the two top levels
        share.append(float(np.mean(t.mv_id[top] == 2)))
    assert share[1] > 2.0 * share[0] and share[1] > 0.5
    t0 = tp.prepare(y, coords, mv)                          # the default is mvbias = 0
    assert np.array_equal(t0.blocking, tp.prepare(y, coords, mv, mvbias=0.0).blocking)


def test_nearest_row_more_exact_ties_than_the_first_candidate_batch():
    """with q >= 3 co-located outcomes and the all-margins fallback (the query's margin has no target) a grid point
    has 4 equidistant neighbours x 3 outcomes = 12 exact ties -- more than the 9 candidates the host path first asks the
    k-d tree for.  The rule (lowest target index among ALL ties, what the device kernel does) must still hold."""
    g = np.stack(np.meshgrid(np.arange(5.0), np.arange(5.0), indexing="ij"), -1).reshape(-1, 2)
    centre = np.array([[2.0, 2.0]])
    nb = g[np.abs(g - centre).sum(axis=1) == 1.0]                  # the 4 neighbours at distance 1
    far = g[np.abs(g - centre).sum(axis=1) > 1.0]
    rng = np.random.default_rng(0)
    tc = np.concatenate([far, np.tile(nb, (3, 1))])                # neighbours of three margins, co-located
    tmv = np.concatenate([rng.integers(0, 3, size=far.shape[0]), np.repeat([0, 1, 2], 4)])
    perm = rng.permutation(tc.shape[0])
    tc, tmv = tc[perm], tmv[perm]
    tmv3 = tmv.copy()                                              # margin 3 is absent among the targets -> all margins compete
    got = tp._nearest_rows(tc, tmv3, centre, np.array([3]), 4, True, None)
    d2 = ((tc - centre) ** 2).sum(axis=1)
    ties = np.nonzero(d2 == d2.min())[0]
    assert ties.size == 12 and got[0] == ties.min()
