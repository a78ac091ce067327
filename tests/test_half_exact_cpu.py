"""The half-float cells of the packed SmithWaterman kernels (csrc/sw_packed.hip PH_PKF_ROW, csrc/sw_traceback.hip
PH_TBF_CELL), restated with numpy float16 on the CPU: under the library's condition -- smax * min(lenA, lenB) <= 2047 and
smax + |gap| <= 2048 -- the recurrence on halves scaled by 2^-11 gives the integer recurrence's H in every cell, and the
direction bits told from the gap-decayed values equal the reference's (align.go:186-227) wherever the walk can read them.
No GPU: this pins the exactness argument the kernels rely on, including the limits."""
import numpy as np
import pytest

S = np.float16(1.0 / 2048.0)


def h16(v):
    x = np.float16(np.float32(v) / np.float32(2048.0))
    assert float(x) * 2048.0 == float(v), v  # the integers the kernels feed are representable
    return x


def clamp01(x):
    return np.float16(min(max(float(x), 0.0), 1.0))


def add(a, b):  # v_pk_add_f16: round to nearest even, like numpy
    return np.float16(a + b)


def int_dp(a, b, mat, gap):
    la, lb = len(a), len(b)
    H = np.zeros((la + 1, lb + 1), np.int64)
    G = np.zeros((la + 1, lb + 1), bool)
    L = np.zeros((la + 1, lb + 1), bool)
    for i in range(1, la + 1):
        for j in range(1, lb + 1):
            d0 = max(H[i - 1, j - 1] + mat[a[i - 1]][b[j - 1]], 0)
            t = max(H[i - 1, j], H[i, j - 1]) + gap
            H[i, j] = max(d0, t)
            G[i, j] = t > d0                   # the gap move wins (the diagonal wins ties, align.go:215)
            L[i, j] = H[i, j - 1] > H[i - 1, j]  # "up" is tested first (align.go:220)
    return H, G, L


def half_dp(a, b, mat, gap, biased):
    """biased: the score pass's row (the value left of a block's first column arrives with the gap subtracted and
    unclamped, the profile's column 0 carries + |gap|); here every column is treated as such a 'column 0'."""
    la, lb = len(a), len(b)
    g = h16(gap)
    H = np.zeros((la + 1, lb + 1), np.float16)
    Gd = np.zeros((la + 1, lb + 1), np.float16)  # clamp(H + gap): what the cell below / right takes
    Gb = np.zeros((la + 1, lb + 1), bool)
    Lb = np.zeros((la + 1, lb + 1), bool)
    for i in range(1, la + 1):
        for j in range(1, lb + 1):
            s = mat[a[i - 1]][b[j - 1]]
            if biased:
                diag_u = add(H[i - 1, j - 1], g)          # unclamped, may be negative
                t = add(diag_u, h16(s - gap))              # profile entry = score + |gap|
            else:
                t = add(H[i - 1, j - 1], h16(s))
            up, left = Gd[i - 1, j], Gd[i, j - 1]
            h = np.float16(max(float(t), float(up), float(left)))  # v_pk_maximum3_f16
            H[i, j] = h
            Gd[i, j] = clamp01(add(h, g))
            Gb[i, j] = float(clamp01(np.float16(h - t))) != 0.0
            Lb[i, j] = float(clamp01(np.float16(left - up))) != 0.0
    return H, Gb, Lb


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("biased", [False, True])
def test_half_recurrence_equals_integer(seed, biased):
    rng = np.random.default_rng(seed)
    ncodes = 4
    smax = int(rng.integers(1, 14))
    mat = rng.integers(-9, smax + 1, (ncodes, ncodes))
    mat[0, 0] = smax
    gap = -int(rng.integers(1, min(40, 2048 - smax) + 1))
    la = int(rng.integers(1, min(60, 2047 // smax) + 1))
    lb = int(rng.integers(1, 90))
    b = rng.integers(0, ncodes, lb)
    a = rng.integers(0, ncodes, la)
    if rng.random() < 0.7:  # a read that aligns well: high scores, long paths
        p = int(rng.integers(0, max(1, lb - la + 1)))
        a[: min(la, lb - p)] = b[p:p + min(la, lb - p)]
        hit = rng.random(la) < 0.08
        a[hit] = rng.integers(0, ncodes, int(hit.sum()))
    H, G, L = int_dp(a, b, mat, gap)
    H16, G16, L16 = half_dp(a, b, mat, gap, biased)
    assert (np.round(H16.astype(np.float64) * 2048.0).astype(np.int64) == H).all()
    assert (H16.astype(np.float64) * 2048.0 == H).all()  # exactly, not after rounding
    live = H > 0                       # the walk stops at 0: bits of other cells are never read
    assert (G16[live] == G[live]).all()
    gapmove = live & G                 # L is consulted only where the gap move won
    assert (L16[gapmove] == L[gapmove]).all()


def test_limits_of_the_condition():
    # the largest score a batch under the condition can reach: smax * len = 2047 (13 * 157 = 2041, 1 * 2047)
    for smax, n in ((13, 157), (1, 2047), (89, 23)):
        a = np.zeros(n, np.int64)
        mat = np.full((2, 2), -3)
        mat[0, 0] = smax
        H16, _, _ = half_dp(a[:40], a[:40], mat, -(2048 - smax), True)  # |gap| at its limit: smax + |gap| = 2048
        H, _, _ = int_dp(a[:40], a[:40], mat, -(2048 - smax))
        assert (H16.astype(np.float64) * 2048.0 == H).all()
        # the diagonal of a perfect match, all the way up: exact at every step
        acc = np.float16(0)
        for k in range(1, n + 1):
            acc = add(acc, h16(smax))
            assert float(acc) * 2048.0 == smax * k
    # one step beyond is NOT representable: 2049 / 2048 rounds (why the library falls back to the int16 cell there)
    assert float(np.float16(np.float32(2049) / np.float32(2048))) * 2048.0 != 2049.0


def _banded_schedule(a, b, mat, gap, RB, nbands):
    """The half-float traceback kernels' SCHEDULE restated in plain integers (csrc/sw_traceback.hip: tb_prof16_kernel = two
    bands in one lane, tb_prof16x2_kernel = four bands in two lanes): band k holds rows [k RB, (k + 1) RB) and works on the
    4-column block tt - k in iteration tt; what it finds above its first row -- four values, their gap-decayed copies, one
    diagonal value -- is what band k - 1 left behind ONE ITERATION EARLIER (the diagonal value: TWO earlier).  Blocks before
    the first and behind the last are pad blocks (score -128), rows behind the read pad rows."""
    la, lb = len(a), len(b)
    nblk = (lb + 3) // 4
    PAD = -128
    out = np.zeros((la + 1, lb + 1), np.int64)
    Hrow = [[0] * RB for _ in range(nbands)]
    in_pr = [[0] * 4 for _ in range(nbands)]
    in_pg = [[0] * 4 for _ in range(nbands)]
    in_pd = [0] * nbands
    for tt in range(nblk + nbands - 1):
        nxt_pr = [row[:] for row in in_pr]
        nxt_pg = [row[:] for row in in_pg]
        nxt_pd = in_pd[:]
        for band in range(nbands):
            bt = tt - band
            real_block = 0 <= bt < nblk
            pr, pg, pdiag = in_pr[band][:], in_pg[band][:], in_pd[band]
            for r in range(RB):
                gi = band * RB + r
                left = Hrow[band][r]
                gl = max(left + gap, 0)
                h, g = [0] * 4, [0] * 4
                for c in range(4):
                    j = 4 * bt + c
                    s = mat[a[gi]][b[j]] if (real_block and gi < la and j < lb) else PAD
                    t = (pdiag if c == 0 else pr[c - 1]) + s
                    h[c] = max(t, pg[c], gl if c == 0 else g[c - 1])  # v_pk_maximum3_f16 (the clamped operands make it >= 0)
                    g[c] = max(h[c] + gap, 0)
                    if real_block and gi < la and j < lb:
                        out[gi + 1, j + 1] = h[c]
                pdiag, pr, pg = left, h, g
                Hrow[band][r] = h[3]
            if band + 1 < nbands:  # band + 1 finds this above its first row in the NEXT iteration
                nxt_pr[band + 1], nxt_pg[band + 1] = pr, pg
                nxt_pd[band + 1] = in_pr[band + 1][3]  # `hd = hh3`: the value that was this iteration's fourth "row above"
        in_pr, in_pg, in_pd = nxt_pr, nxt_pg, nxt_pd
    return out


@pytest.mark.parametrize("RB,nbands", [(8, 2), (8, 4), (16, 4)])
def test_the_band_schedule_reproduces_the_recurrence(RB, nbands):
    rng = np.random.default_rng(RB * 10 + nbands)
    mat = [[0, 0, 0, 0, 0], [0, 5, -4, -4, -4], [0, -4, 5, -4, -4], [0, -4, -4, 5, -4], [0, -4, -4, -4, 5]]
    for it in range(25):
        la = int(rng.integers(1, RB * nbands + 1))
        lb = int(rng.integers(1, 70))
        b = rng.integers(1, 5, lb)
        a = rng.integers(1, 5, la)
        if it % 2 == 0 and lb >= 4:  # related: long positive paths through every band
            p = int(rng.integers(0, lb))
            a = np.resize(np.concatenate([b[p:], b[:p]]), la).copy()
            hit = rng.random(la) < 0.1
            a[hit] = rng.integers(1, 5, int(hit.sum()))
        gap = -int(rng.integers(1, 6))
        H, _, _ = int_dp(a, b, mat, gap)
        got = _banded_schedule(a, b, mat, gap, RB, nbands)
        assert (got == H).all(), (it, la, lb, gap)
