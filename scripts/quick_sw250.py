"""250-bp reads against one 5 kb reference: the two-lanes-per-pair half-float traceback (path 5) beside the
one-wave-per-pair kernel (POLYHIP_TB_HALF2=0, path 4), and the oracle on a sample of the fused call's output."""
import os
import sys

import torch

sys.path.insert(0, '.')
import oracle as orc  # noqa: E402  (a script, not the product)
from poly_amd import bench_extra  # noqa: E402

dev = torch.device('cuda:0')
KEYS = ('score_pass_ms', 'traceback_ms', 'align_one_call_ms', 'cell_updates_per_s', 'cell_updates_per_s_align_one_call',
        'score_path', 'traceback_path', 'mean_alignment_len')
om = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
for mode in ('1', '0'):
    os.environ['POLYHIP_TB_HALF2'] = mode
    for n, L in ((400_000, 250), (400_000, 200)) + (((1_000_000, 150),) if mode == '1' else ()):
        r = bench_extra.sw(dev, n, L)
        print('HALF2=' + mode, n, L, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in KEYS}, flush=True)
        sp = r['_spot']
        ok = True
        for pr in sp['pairs']:
            s, sa, sb, ea, eb = orc.smith_waterman(pr['read'], sp['ref'], om, sp['gap'])
            sa = sa if isinstance(sa, bytes) else sa.encode()
            sb = sb if isinstance(sb, bytes) else sb.encode()
            ok &= (pr['score'], pr['endA'], pr['endB'], pr['alnA'], pr['alnB']) == (s, ea, eb, sa, sb)
        print('   oracle spot check:', ok, flush=True)
