"""The traceback's per-pair window (csrc/sw_traceback.hip, pair_window), restated and checked without a GPU.

The kernels re-run the Smith-Waterman recurrence (align.go:171-203) on `need` columns ending at the end cell and walk the
direction bits (align.go:205-231).  Round 4 shortened the window from  span + eA + over  to  span + over  columns
(span = eA + lw, lw = (smax*eA - M)/|gap|, over = (smax*eA + smax - M)/|gap| + 1): the bound on how far left a candidate's
optimal path can begin is taken per walk cell (a cell in row i sits at column >= eB - (eA - i) - lw and a path into one
of its candidates climbs at most i rows), not once for the leftmost walk cell and the bottom row together.

Here: the full-matrix walk of the reference against the walk over a matrix computed ONLY inside the window (zero boundary
left of it), for random scorings, related and unrelated pairs -- every step's decision must be the same.  Also: the bound
is tight enough to matter (on config-4-like reads it is ~200 columns, the old sum ~350)."""
import numpy as np


def pair_window(wcols, eA, M, smax, gap, wide):
    """csrc/sw_traceback.hip pair_window, line by line"""
    if gap >= 0 or smax <= 0 or M <= 0:
        return wcols
    g, top = -gap, smax * eA
    span = eA + ((top - M) // g if top > M else 0)
    over = (top + smax - M) // g + 1 if top + smax > M else 0
    need = (span + eA + top // g if wide else span + min(over, top // g)) + 2
    return min(need, wcols)


def sw_matrix(a, b, S, gap, first_col=1):
    """H over columns first_col..len(b) (1-based); everything left of first_col is a zero boundary"""
    H = np.zeros((len(a) + 1, len(b) + 1), np.int64)
    for i in range(1, len(a) + 1):
        for j in range(first_col, len(b) + 1):
            H[i, j] = max(0, H[i - 1, j - 1] + S[a[i - 1], b[j - 1]], H[i - 1, j] + gap, H[i, j - 1] + gap)
    return H


def walk(H, a, b, S, gap, i, j, stop_col=0):
    """align.go:205-231: diagonal first, then up, then left; the list of moves"""
    moves = []
    while H[i, j] > 0 and i > 0 and j > stop_col:
        if H[i, j] == H[i - 1, j - 1] + S[a[i - 1], b[j - 1]]:
            moves.append("d")
            i, j = i - 1, j - 1
        elif H[i, j] == H[i - 1, j] + gap:
            moves.append("u")
            i -= 1
        else:
            moves.append("l")
            j -= 1
    return moves


def _case(rng, related):
    nsym = int(rng.integers(2, 5))
    style = int(rng.integers(0, 2))
    if style == 0:
        S = rng.integers(-6, 7, (nsym, nsym))
    else:
        S = np.full((nsym, nsym), -int(rng.integers(0, 6)))
        np.fill_diagonal(S, int(rng.integers(1, 8)))
    gap = -int(rng.integers(1, 6))
    lb = int(rng.integers(20, 220))
    b = rng.integers(0, nsym, lb)
    la = int(rng.integers(1, 40))
    if related:
        p = int(rng.integers(0, max(1, lb - la)))
        a = b[p:p + la].copy()
        hit = rng.random(len(a)) < rng.random() * 0.4
        a[hit] = rng.integers(0, nsym, int(hit.sum()))
        if rng.random() < 0.5 and len(a) > 3:  # an indel
            c = int(rng.integers(1, len(a) - 1))
            a = np.concatenate([a[:c], a[c + 1:], a[:1]])
    else:
        a = rng.integers(0, nsym, la)
    return a, b, S, gap


def test_windowed_walk_equals_the_full_matrix_walk():
    rng = np.random.default_rng(20260924)
    checked = narrowed = 0
    for it in range(700):
        a, b, S, gap = _case(rng, related=it % 3 != 0)
        smax = int(S.max())
        H = sw_matrix(a, b, S, gap)
        M = int(H.max())
        if M <= 0:
            continue
        eA, eB = (int(x) for x in np.argwhere(H == M)[0])  # row-major first maximum (align.go:197)
        want = walk(H, a, b, S, gap, eA, eB)
        wcols = len(b)
        for wide in (0, 1):
            need = pair_window(wcols, eA, M, smax, gap, wide)
            # the kernels: c_s = eB - mycols + 1 (first column of the DP, 1-based), rounded down to a block of four
            c_s = eB - need + 1 if eB > need else 1
            jb0 = (c_s - 1) & ~3
            Hw = sw_matrix(a, b, S, gap, first_col=jb0 + 1)
            assert (Hw <= H).all()
            got = walk(Hw, a, b, S, gap, eA, eB, stop_col=jb0)
            assert got == want, (it, wide, eA, eB, M, smax, gap, need)
        checked += 1
        narrowed += pair_window(wcols, eA, M, smax, gap, 0) < pair_window(wcols, eA, M, smax, gap, 1)
    assert checked > 400 and narrowed > 100


def test_the_batch_wide_window_covers_every_score():
    # k3t::window: wcols = lenA + 2 * (smax * lenA / |gap|) + 2 must cover pair_window's need for every end row and score >= 1
    for smax, gap, lenA in ((5, -2, 150), (5, -7, 150), (11, -1, 64), (1, -9, 256), (3, -3, 37)):
        wcols = lenA + 2 * ((smax * lenA) // -gap) + 2
        for eA in range(1, lenA + 1):
            for M in range(1, smax * eA + 1):
                assert pair_window(10 ** 9, eA, M, smax, gap, 0) <= wcols, (smax, gap, lenA, eA, M)


def test_the_bound_on_a_config4_read():
    # 150 rows, NUC_4 (smax 5), gap -2, a read that aligns with M = 700 of 750: the old sum was 355 columns
    assert pair_window(902, 150, 700, 5, -2, 0) == 150 + 25 + 28 + 2
    assert pair_window(10 ** 9, 150, 700, 5, -2, 1) == 150 + 25 + 150 + 375 + 2
    # nothing known about the score: the batch-wide window
    assert pair_window(902, 150, 0, 5, -2, 0) == 902
    # a positive gap score, a matrix without a positive entry: the whole of B
    assert pair_window(300, 40, 17, 0, 1, 0) == 300 and pair_window(300, 40, 17, 5, 0, 0) == 300


def test_window_of_a_deferred_end_cell_covers_the_pairs_own_window():
    """Round 5, reads of 257..1024 rows in one call: the score pass leaves (M, the last column jend of the only 4-column block
    that holds M) and the one-wave-per-pair traceback sweeps  pair_window(lenA) + 4  columns ending at jend, not knowing the
    end row yet.  That window covers what the kernel would take knowing the end cell (eA <= lenA, jend - 3 <= eB <= jend):
    pair_window grows with the row count, in both of its forms.  And it covers the locate kernel's own window
    (lenA + (smax * lenA - M) / |gap| + 4 columns ending at jend), in which the first cell worth M is found."""
    rng = np.random.default_rng(12)
    for it in range(20000):
        smax = int(rng.integers(1, 40))
        gap = -int(rng.integers(1, 30))
        lenA = int(rng.integers(1, 1025))
        eA = int(rng.integers(1, lenA + 1))
        M = int(rng.integers(1, smax * eA + 1))  # the end cell is in row eA: M <= smax * eA
        wcols = 10 ** 9
        for wide in (0, 1):
            w_known = pair_window(wcols, eA, M, smax, gap, wide)
            w_defer = pair_window(wcols, lenA, M, smax, gap, wide) + 4
            assert pair_window(wcols, eA, M, smax, gap, wide) <= pair_window(wcols, min(lenA, eA + 1), M, smax, gap, wide)
            for back in range(4):  # eB = jend - back
                assert w_defer >= w_known + back
        g, top = -gap, smax * lenA
        locate_need = lenA + ((top - M) // g if top > M else 0) + 4
        assert pair_window(wcols, lenA, M, smax, gap, 0) + 4 >= locate_need
