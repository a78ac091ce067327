"""Traceback ablation: the half-float byte-profile kernels with and without their walk (POLYHIP_TB_NOWALK=1: empty
strings), config 4 and 400k x 250 bp; stand-alone traceback, score known."""
import os
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import align, alphabet, matrix, workloads
from poly_amd.bench_extra import _time
dev = torch.device('cuda:0')
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
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
    for tag, env in (("walk", {}), ("no walk", {"POLYHIP_TB_NOWALK": "1"}), ("no overlap", {"POLYHIP_TB_OVERLAP": "0"}),
                     ("no overlap, no walk", {"POLYHIP_TB_OVERLAP": "0", "POLYHIP_TB_NOWALK": "1"})):
        for k in ("POLYHIP_TB_NOWALK", "POLYHIP_TB_OVERLAP"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ms = _time(lambda: align.sw_traceback_dev(sc, A, offA, LA, B, None, LB, ea, eb, er, alnA, alnB, ln, tbw, score_t=score), 5)
        print(f"{n} x {LA}: {tag}: {ms:.2f} ms  path {align.sw_traceback_last_path()}  mean len {float(ln.double().mean()):.1f}", flush=True)
    for k in ("POLYHIP_TB_NOWALK", "POLYHIP_TB_OVERLAP"):
        os.environ.pop(k, None)
