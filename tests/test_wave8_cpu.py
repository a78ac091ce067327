"""CPU restatements of the round-5 long-read Smith-Waterman kernels (csrc/sw_traceback.hip tb_wave_kernel<R, true>,
csrc/sw_wave.hip sw_wave8_kernel): the byte-profile cell in integers against the recurrence of align.go:171-231, and the
window logic of the walk that runs out of the wave's registers.  No GPU, no library: what is checked here is the
arithmetic and the index logic the kernels rely on; the kernels themselves are compared with the oracle in
tests/test_traceback_gpu.py and tests/test_align_gpu.py."""
import numpy as np


def _reference(a, b, S, gap):
    """H and the traceback's decision per cell as align.go:185-231 makes it: 0 = diagonal, 1 = up, 2 = left"""
    n, m = len(a), len(b)
    H = np.zeros((n + 1, m + 1), np.int64)
    D = np.zeros((n + 1, m + 1), np.int8)
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            d = H[i - 1, j - 1] + S[a[i - 1], b[j - 1]]
            u = H[i - 1, j] + gap
            l = H[i, j - 1] + gap
            h = max(0, d, u, l)
            H[i, j] = h
            D[i, j] = 0 if h == d else 1 if h == u else 2  # :215-227: diagonal first, then up, else left
    return H, D


def test_byte_profile_cell_equals_the_recurrence_and_its_decisions():
    """x = diag' + (s - gap); t = max(up', left'); h' = max3(x, t, 0) + gap with every H kept as H + gap: h' - gap is the
    recurrence's H in every cell, and on every cell the walk can visit (H > 0) the two bits -- G = t > x, L = left' > up' --
    are the reference's decision.  (G compares with x, not with max(x, 0): the test shows the two differ only where H = 0.)
    Matrices at both ends of the byte range, gaps 1..120, alphabets of 2..6 symbols."""
    rng = np.random.default_rng(8)
    differ_at_zero = 0
    for it in range(60):
        nsym = int(rng.integers(2, 7))
        gap = -int(rng.integers(1, 12)) if it % 7 else -int(rng.integers(100, 121))  # (a positive score must stay possible)
        hi, lo = 127 + gap, -128 + gap  # s - gap must fit a signed byte
        S = rng.integers(max(lo, -40), min(hi, 40) + 1, (nsym, nsym))
        S[0, 0] = hi
        S[nsym - 1, 0] = lo
        S[np.arange(nsym), np.arange(nsym)] = np.maximum(S[np.arange(nsym), np.arange(nsym)], 1)
        S[0, 0] = hi
        n, m = int(rng.integers(1, 40)), int(rng.integers(1, 60))
        a, b = rng.integers(0, nsym, n), rng.integers(0, nsym, m)
        if it % 3 == 0:  # a read that is a mutated window of the reference: long positive runs
            m = max(m, n)
            b = rng.integers(0, nsym, m)
            a = b[:n].copy()
            a[rng.random(n) < 0.1] = rng.integers(0, nsym)
        H, D = _reference(a, b, S, gap)
        prof = (S - gap).astype(np.int64)  # the byte the plane holds
        assert prof.min() >= -128 and prof.max() <= 127
        lg = np.full(n + 1, gap, np.int64)  # H + gap of the previous column, row 0 = the boundary (H = 0)
        for j in range(1, m + 1):
            new = np.full(n + 1, gap, np.int64)
            for i in range(1, n + 1):
                diag, up, left = lg[i - 1], new[i - 1], lg[i]
                x = diag + prof[a[i - 1], b[j - 1]]
                t = max(up, left)
                hg = max(x, t, 0) + gap
                new[i] = hg
                assert hg - gap == H[i, j]
                G, L = t > x, left > up
                G_clamped = t > max(x, 0) + 0  # what the table form pushes: t + ... > max(diag + s, 0)
                if H[i, j] > 0:
                    assert G == G_clamped
                    want = D[i, j]
                    assert (0 if not G else 1 if not L else 2) == want, (it, i, j)
                elif G != G_clamped:
                    differ_at_zero += 1
            lg = new
    assert differ_at_zero > 0  # the shortcut is a real one, and harmless


def test_locate_question_finds_the_row_major_first_maximum():
    """sw_wave8_kernel / the deferred end cell of tb_wave_kernel<R, true>: per column, is the largest h' of a lane's rows
    M + gap?  The smallest row that ever says yes, with the first column in which it does, is align.go:197-201's end cell
    (strict >, row-major order); rows behind the read's end carry -128 in the locate kernel's planes and never hold M."""
    rng = np.random.default_rng(9)
    for it in range(40):
        nsym = 4
        S = np.full((nsym, nsym), -4, np.int64)
        S[np.arange(nsym), np.arange(nsym)] = 5
        gap = -2
        n, m = int(rng.integers(2, 30)), int(rng.integers(2, 80))
        b = rng.integers(0, nsym, m)
        if it % 2:  # a tandem repeat: the maximum occurs in several places
            unit = rng.integers(0, nsym, 7)
            b = np.tile(unit, m // 7 + 1)[:m]
        a = b[:n].copy() if n <= m else rng.integers(0, nsym, n)
        a[rng.random(len(a)) < 0.15] = rng.integers(0, nsym)
        n = len(a)
        H, _ = _reference(a, b, S, gap)
        M = int(H.max())
        if M == 0:
            continue
        best = None  # the reference's argmax: first maximum in row-major order
        for i in range(1, n + 1):
            for j in range(1, m + 1):
                if H[i, j] == M and best is None:
                    best = (i, j)
        R = 8
        rows = -(-n // R) * R  # the lanes' rows, pad rows behind the read
        besti, bestj = None, None
        Hpad = np.zeros((rows + 1, m + 1), np.int64)
        Hpad[:n + 1] = H
        for i in range(n + 1, rows + 1):  # a pad row: profile byte -128 -> x far below, h = max(t, 0)
            for j in range(1, m + 1):
                Hpad[i, j] = max(0, max(Hpad[i - 1, j], Hpad[i, j - 1]) + gap)
        assert (Hpad[n + 1:] < M).all()
        for j in range(1, m + 1):  # columns in order (every lane sees them in order)
            for lane in range(rows // R):
                col = Hpad[lane * R + 1: lane * R + R + 1, j]
                if col.max() == M:  # the question; rarely yes
                    for k in range(R):
                        r = lane * R + k
                        if col[k] == M and r < n and (besti is None or r < besti):
                            besti, bestj = r, j
        assert (besti + 1, bestj) == best


def _walk_windows(R, SP, moves, i0, j0, c_s):
    """the walk's window logic (tb_wave_kernel: `lc`, `wtop0`, `wtop1`, lane q holds word wtop[q >> 5] - (q & 31) of lane row
    lc - (q >> 5)); returns the number of fetches.  Asserts that every hit reads the word it asks for."""
    lanes = [None] * 64
    lc, wtop0, wtop1 = None, 0, 0
    i, j, fetches, steps = i0, j0, 0, 0
    for mv in moves:
        if i == 0 or j < c_s:
            break
        while True:
            r = i - 1
            l, k = divmod(r, R)
            s = (j - c_s) + l
            wi = s // SP
            m = None if lc is None else lc - l
            top = wtop0 if m == 0 else wtop1
            if m in (0, 1) and wi <= top and top - wi < 32:
                idx = m * 32 + (top - wi)
                assert 0 <= idx < 64 and lanes[idx] == (wi, l), (lanes[idx], wi, l)
                break
            lc, wtop0, wtop1 = l, wi, ((s - 1) // SP if s else 0)
            for q in range(64):
                mq, x = q >> 5, q & 31
                wt = wtop1 if mq else wtop0
                ok = l >= mq and x <= wt and (mq == 0 or s > 0)
                lanes[q] = (wt - x, l - mq) if ok else None
            fetches += 1
        steps += 1
        if mv == 0:
            i, j = i - 1, j - 1
        elif mv == 1:
            i -= 1
        else:
            j -= 1
    return fetches, steps


def test_walk_windows_always_hold_the_word_a_step_asks_for():
    """every (rows per lane, steps per word) the kernels use; diagonal walks, walks with runs of gaps either way, walks that
    hug the window's first column; a fetch per ~30 diagonal steps, never an endless refetch"""
    rng = np.random.default_rng(10)
    for R, SP in ((4, 4), (8, 2), (16, 1)):
        for it in range(150):
            n = int(rng.integers(1, 64 * R + 1))
            c_s = int(rng.integers(1, 50))
            width = int(rng.integers(1, 1500))
            j0 = c_s + width - 1
            pg = [0.0, 0.02, 0.3, 0.9][it % 4]
            moves = []
            while len(moves) < 4000:
                if rng.random() < pg:
                    moves += [int(rng.integers(1, 3))] * int(rng.integers(1, 40))
                else:
                    moves.append(0)
            fetches, steps = _walk_windows(R, SP, moves, n, j0, c_s)
            assert steps > 0 and fetches <= steps + 1
            if pg == 0.0 and steps >= 200:
                assert fetches <= steps // (14 if SP == 1 else 7) + 2, (R, SP, fetches, steps)


def test_integers_below_0x7c00_order_like_halves():
    """csrc/sw_packed.hip PH_PK_ROW (round 5): the int16 packed kernels take their block maximum with v_pk_maximum3_f16.  That is
    an integer maximum as long as every operand is in [0, 0x7C00): non-negative halves -- denormals included, the kernels run
    with them preserved -- are strictly increasing in their bit patterns, and 0x7C00 (infinity) / NaNs are never reached
    (packed_plan: smax * min(lenA, lenB) < 30000).  The cell itself cannot use it: diag + score can be -1 .. -128, and those
    16-bit patterns are NaNs, which an IEEE-754-2019 maximum propagates."""
    bits = np.arange(0, 0x7C00, dtype=np.uint16)
    h = bits.view(np.float16).astype(np.float64)
    assert np.isfinite(h).all() and (np.diff(h) > 0).all() and h[0] == 0.0
    assert 30000 < 0x7C00
    rng = np.random.default_rng(11)
    a, b, c = (rng.integers(0, 30000, 100_000).astype(np.uint16) for _ in range(3))
    as_half = np.maximum(np.maximum(a.view(np.float16), b.view(np.float16)), c.view(np.float16)).view(np.uint16)
    assert (as_half == np.maximum(np.maximum(a, b), c)).all()
    neg = np.arange(-128, 0, dtype=np.int16).view(np.uint16).view(np.float16)
    assert np.isnan(neg.astype(np.float64)).all()
