"""Randomised cross-check of the Smith-Waterman paths (not part of the test suite; run on the GPU box):
random substitution matrices / gaps / lengths / alphabets; every pair: packed pass (3) == exact lane-per-pair
kernel (1) == wave kernel (4, on a slice); traceback: profile kernel == table kernel; a sample == oracle."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
import oracle as orc
from poly_amd import align, alphabet, matrix
dev = torch.device('cuda:0')
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + it)
    it += 1
    nsym = int(rng.integers(2, 7))
    syms = "ACGTNR"[:nsym]
    style = int(rng.integers(0, 3))
    if style == 0:
        mat = rng.integers(-9, 10, (nsym, nsym))
    elif style == 1:
        ma, mi = int(rng.integers(1, 12)), -int(rng.integers(0, 12))
        mat = np.full((nsym, nsym), mi); np.fill_diagonal(mat, ma)
    else:
        mat = rng.integers(-3, 4, (nsym, nsym))
    mat = mat.astype(int).tolist()
    gap = -int(rng.integers(1, 10))
    a = alphabet.NewAlphabet(list(syms))
    sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, mat), gap)
    om = orc.SubstitutionMatrix(syms, syms, mat)
    LB = int(rng.choice([7, 60, 333, 1000, 5000, 9000]))
    L = int(rng.choice([5, 40, 64, 100, 150, 152, 153, 200, 256, 257, 300, 500, 700, 1100, 1300, 2048]))
    n = 50_000
    if L > 256:  # long reads: enough pairs for the packed banded pass (path 7), score pass only
        rows = next(r for r in (304, 512, 608, 1024, 1216, 2048) if r >= L)
        n = (8 << 20) // rows + 7
    symb = np.frombuffer(syms.encode(), np.uint8)
    ref = symb[rng.integers(0, nsym, LB)]
    if rng.random() < 0.3:  # repeats: many ties
        unit = ref[: max(3, LB // 9)]
        ref = np.tile(unit, LB // len(unit) + 1)[:LB].copy()
    starts = rng.integers(0, max(1, LB - L + 1), n)
    reads = ref[(starts[:, None] + np.arange(L)[None, :]) % LB]
    rate = rng.random(n)[:, None] * float(rng.choice([0.1, 0.5, 1.0]))
    hit = rng.random((n, L)) < rate
    reads[hit] = symb[rng.integers(0, nsym, int(hit.sum()))]
    lens = rng.integers(0, L + 1, n)
    lens[rng.random(n) < 0.6] = L
    offs = np.zeros(n + 1, np.int64); offs[1:] = np.cumsum(lens)
    flat = np.concatenate([reads[i, :lens[i]] for i in range(n)])
    A = torch.from_numpy(flat.copy()).to(dev); offA = torch.from_numpy(offs).to(dev)
    B = torch.from_numpy(ref.copy()).to(dev)

    def score_pass(env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            score = torch.full((n,), -7, dtype=torch.int64, device=dev)
            ea, eb, er = (torch.full((n,), -7, dtype=torch.int32, device=dev) for _ in range(3))
            work = torch.empty(align.sw_workspace_bytes(sc, n, L, LB), dtype=torch.uint8, device=dev)
            align.sw_batch_dev(sc, A, offA, L, B, None, LB, score, ea, eb, er, work)
            torch.cuda.synchronize()
            return (score, ea, eb, er), align.last_path()
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)

    got3, p3 = score_pass({})
    got1, p1 = score_pass({"POLYHIP_SW_PACKED": "0"})
    for x, y in zip(got3, got1):
        assert torch.equal(x, y), ("packed vs exact", it, syms, mat, gap, LB, L, p3, p1)
    if L > 256:
        s_h, ea_h, eb_h = (t.cpu().numpy() for t in got3[:3])
        refb = ref.tobytes()
        for p in range(0, n, max(1, n // 6)):
            rd = flat[offs[p]:offs[p + 1]].tobytes()
            s, _, _, oa, ob = orc.smith_waterman(rd, refb, om, gap)
            assert (int(s_h[p]), int(ea_h[p]), int(eb_h[p])) == (s, oa, ob), ("oracle score", it, p, syms, mat, gap, LB, L)
        print(f"it {it}: syms {syms} gap {gap} LB {LB} L {L} n {n} paths {p3}/{p1} max score {int(got3[0].max())} ok", flush=True)
        continue
    # wave kernel on the first 3000 pairs
    m = 3000
    sw = torch.full((m,), -7, dtype=torch.int64, device=dev)
    wa, wb, we = (torch.full((m,), -7, dtype=torch.int32, device=dev) for _ in range(3))
    work = torch.empty(align.sw_workspace_bytes(sc, m, L, LB), dtype=torch.uint8, device=dev)
    align.sw_batch_dev(sc, A, offA[: m + 1].contiguous(), L, B, None, LB, sw, wa, wb, we, work)
    torch.cuda.synchronize()
    p4 = align.last_path()
    for x, y in zip((sw, wa, wb, we), got1):
        assert torch.equal(x, y[:m]), ("wave vs exact", it, syms, mat, gap, LB, L, p4)
    # traceback: profile kernel (score given) vs table kernel (no score)
    score, ea, eb, er = got1
    stride = align.sw_traceback_stride(sc, L, LB)
    tbw = torch.empty(min(align.sw_traceback_workspace_bytes(sc, n, L, LB), 2 << 30), dtype=torch.uint8, device=dev)
    outs = []
    for st in (score, None):
        aa = torch.zeros((n, stride), dtype=torch.uint8, device=dev); bb = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        ln = torch.zeros(n, dtype=torch.int32, device=dev)
        align.sw_traceback_dev(sc, A, offA, L, B, None, LB, ea, eb, er, aa, bb, ln, tbw, score_t=st)
        torch.cuda.synchronize()
        outs.append((aa, bb, ln, align.sw_traceback_last_path()))
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), \
        ("traceback kernels", it, syms, mat, gap, LB, L, outs[0][3], outs[1][3])
    # oracle sample
    s_h, ea_h, eb_h = score.cpu().numpy(), ea.cpu().numpy(), eb.cpu().numpy()
    a_h, b_h, l_h = outs[0][0].cpu().numpy(), outs[0][1].cpu().numpy(), outs[0][2].cpu().numpy()
    refb = ref.tobytes()
    step = max(1, n // max(4, int(2e6 // max(1, L * LB))))
    for p in range(0, n, step):
        rd = flat[offs[p]:offs[p + 1]].tobytes()
        s, sa, sb, oa, ob = orc.smith_waterman(rd, refb, om, gap)
        sa = sa if isinstance(sa, bytes) else sa.encode(); sb = sb if isinstance(sb, bytes) else sb.encode()
        assert (int(s_h[p]), int(ea_h[p]), int(eb_h[p])) == (s, oa, ob), ("oracle score", it, p, syms, mat, gap, LB, L)
        assert a_h[p, stride - l_h[p]:].tobytes() == sa and b_h[p, stride - l_h[p]:].tobytes() == sb, ("oracle strings", it, p, syms, mat, gap, LB, L)
    # per-pair B: register-tiled kernel (5) vs generic (2), any gap sign
    npp = 4000
    gap2 = int(rng.integers(-9, 3))
    sc2 = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, mat), gap2)
    LBp = int(rng.choice([1, 9, 70, 200, 300]))
    lb = rng.integers(0, LBp + 1, npp)
    offb = np.zeros(npp + 1, np.int64); offb[1:] = np.cumsum(lb)
    Bp = symb[rng.integers(0, nsym, int(offb[-1]) + 1)]
    # make most B's related to their A: copy a prefix of the read
    for q in range(0, npp, 3):
        w = min(int(lb[q]), int(lens[q]))
        Bp[offb[q]:offb[q] + w] = flat[offs[q]:offs[q] + w]
    Bt = torch.from_numpy(Bp.copy()).to(dev); offBt = torch.from_numpy(offb).to(dev)
    offAp = offA[: npp + 1].contiguous()
    res = []
    for env in ({}, {"POLYHIP_SW_PAIR": "0"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            sco = torch.full((npp,), -7, dtype=torch.int64, device=dev)
            xa, xb, xe = (torch.full((npp,), -7, dtype=torch.int32, device=dev) for _ in range(3))
            wk = torch.empty(align.sw_workspace_bytes(sc2, npp, L, LBp, False), dtype=torch.uint8, device=dev)
            align.sw_batch_dev(sc2, A, offAp, L, Bt, offBt, LBp, sco, xa, xb, xe, wk)
            torch.cuda.synchronize()
            res.append(((sco, xa, xb, xe), align.last_path()))
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    for x, y in zip(res[0][0], res[1][0]):
        assert torch.equal(x, y), ("pair kernel vs generic", it, syms, mat, gap2, LBp, L, res[0][1], res[1][1])
    om2 = orc.SubstitutionMatrix(syms, syms, mat)
    sh, ah, bh = (t.cpu().numpy() for t in res[0][0][:3])
    for q in range(0, npp, 97):
        s_, _, _, oa, ob = orc.smith_waterman(flat[offs[q]:offs[q + 1]].tobytes(), Bp[offb[q]:offb[q + 1]].tobytes(), om2, gap2)
        assert (int(sh[q]), int(ah[q]), int(bh[q])) == (s_, oa, ob), ("pair oracle", it, q, syms, mat, gap2)
    # ... and its traceback: the per-lane-profile half-float kernel (6) where it applies, against the table kernel (2)
    sco, xa, xb, xe = res[0][0]
    strp = align.sw_traceback_stride(sc2, L, LBp)
    tbp = torch.empty(align.sw_traceback_workspace_bytes(sc2, npp, L, LBp), dtype=torch.uint8, device=dev)
    tbo = []
    for env in ({}, {"POLYHIP_TB_PAIR16": "0"}):
        os.environ.update(env)
        pa = torch.zeros((npp, strp), dtype=torch.uint8, device=dev); pb = torch.zeros((npp, strp), dtype=torch.uint8, device=dev)
        pl = torch.zeros(npp, dtype=torch.int32, device=dev)
        align.sw_traceback_dev(sc2, A, offAp, L, Bt, offBt, LBp, xa, xb, xe, pa, pb, pl, tbp, score_t=sco)
        torch.cuda.synchronize()
        tbo.append((pa, pb, pl, align.sw_traceback_last_path()))
        for k in env:
            os.environ.pop(k, None)
    livep = torch.arange(strp, device=dev)[None, :] >= (strp - tbo[0][2].long())[:, None]
    assert torch.equal(tbo[0][2], tbo[1][2]) and bool(((tbo[0][0] == tbo[1][0]) | ~livep).all()) and bool(((tbo[0][1] == tbo[1][1]) | ~livep).all()), \
        ("pair traceback kernels", it, syms, mat, gap2, LBp, L, tbo[0][3], tbo[1][3])
    pah, pbh, plh = tbo[0][0].cpu().numpy(), tbo[0][1].cpu().numpy(), tbo[0][2].cpu().numpy()
    for q in range(0, npp, 197):
        s_, sa_, sb_, _, _ = orc.smith_waterman(flat[offs[q]:offs[q + 1]].tobytes(), Bp[offb[q]:offb[q + 1]].tobytes(), om2, gap2)
        sa_ = sa_ if isinstance(sa_, bytes) else sa_.encode(); sb_ = sb_ if isinstance(sb_, bytes) else sb_.encode()
        assert pah[q, strp - plh[q]:].tobytes() == sa_ and pbh[q, strp - plh[q]:].tobytes() == sb_, \
            ("pair oracle strings", it, q, syms, mat, gap2, tbo[0][3], pah[q, strp - plh[q]:].tobytes(), pbh[q, strp - plh[q]:].tobytes(), sa_, sb_,
             flat[offs[q]:offs[q + 1]].tobytes(), Bp[offb[q]:offb[q + 1]].tobytes(), int(sh[q]), int(ah[q]), int(bh[q]), strp)
    print(f"it {it}: syms {syms} gap {gap} LB {LB} L {L} pp {res[0][1]}/{res[1][1]} tbp {tbo[0][3]}/{tbo[1][3]} gap2 {gap2} paths {p3}/{p1}/{p4} tb {outs[0][3]}/{outs[1][3]} max score {int(score.max())} ok", flush=True)
print("fuzz done", it, "iterations")
