"""Traceback of the half-float byte-profile kernels, config 4 and 400k x 250 bp: the walk inside the sweep kernel (default)
against the walk as a kernel of its own (POLYHIP_TB_SPLITWALK=1) and no walk at all (POLYHIP_TB_NOWALK=1: empty strings);
stand-alone traceback with the score known, and score + strings in one call."""
import os
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import align, alphabet, matrix, workloads
from poly_amd.bench_extra import _time
dev = torch.device('cuda:0')
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
KEYS = ("POLYHIP_TB_NOWALK", "POLYHIP_TB_OVERLAP", "POLYHIP_TB_SPLITWALK")
for n, LA in ((1_000_000, 150), (400_000, 250)):
    LB = 5000
    B, A = workloads.config4_reads(n, LA, LB, first=0, device=dev)
    A = A.reshape(-1).contiguous()
    offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
    score = torch.zeros(n, dtype=torch.int64, device=dev)
    ea, eb, er, ln = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4))
    work = torch.empty(align.sw_workspace_bytes(sc, n, LA, LB, True), dtype=torch.uint8, device=dev)
    align.sw_batch_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, work)
    stride = align.sw_traceback_stride(sc, LA, LB)
    tbw = torch.empty(align.sw_traceback_workspace_bytes(sc, n, LA, LB), dtype=torch.uint8, device=dev)
    alnA = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    alnB = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    ref = None
    for tag, env in (("walk in the sweep kernel", {}), ("walk kernel", {"POLYHIP_TB_SPLITWALK": "1"}),
                     ("no walk", {"POLYHIP_TB_NOWALK": "1"}), ("no overlap", {"POLYHIP_TB_OVERLAP": "0"})):
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        ms = _time(lambda: align.sw_traceback_dev(sc, A, offA, LA, B, None, LB, ea, eb, er, alnA, alnB, ln, tbw, score_t=score), 5)
        same = ""
        if "NOWALK" not in "".join(env):
            if ref is None:
                ref = (alnA.clone(), alnB.clone(), ln.clone())
            else:
                live = torch.arange(stride, device=dev)[None, :] >= (stride - ln.long())[:, None]
                same = "  same strings: %s" % bool(torch.equal(ln, ref[2]) and bool(((alnA == ref[0]) | ~live).all()) and bool(((alnB == ref[1]) | ~live).all()))
        print(f"{n} x {LA}: {tag}: {ms:.2f} ms  path {align.sw_traceback_last_path()}  mean len {float(ln.double().mean()):.1f}{same}", flush=True)
    for k in KEYS:
        os.environ.pop(k, None)
    for tag, env in (("walk in the sweep kernel", {}), ("walk kernel", {"POLYHIP_TB_SPLITWALK": "1"})):
        os.environ.update(env)
        ms = _time(lambda: align.sw_align_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, alnA, alnB, ln, work, tbw), 5)
        print(f"{n} x {LA}: score + strings in one call, {tag}: {ms:.2f} ms = {n * LA * LB / ms * 1e3:.3e} cell updates/s", flush=True)
        for k in KEYS:
            os.environ.pop(k, None)
