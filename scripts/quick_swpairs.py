"""reads against reads (every pair its own B): the per-lane-profile half-float traceback (path 6) against the table kernel
(POLYHIP_TB_PAIR16=0, path 2); 200k pairs of 150 x 150 and 100 x 100."""
import os
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import align, bench_extra
dev = torch.device('cuda:0')
for mode in ('1', '0'):
    os.environ['POLYHIP_TB_PAIR16'] = mode
    for n, L in ((200_000, 150), (400_000, 100)):
        r = bench_extra.sw_pairs(dev, n, L)
        print('PAIR16=' + mode, n, L, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != 'workload'},
              'tb path', align.sw_traceback_last_path(), flush=True)
