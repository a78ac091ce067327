"""The small legs of bench.py for the profiler, each launched exactly CALLS times (bench_extra._time replaced), in two
groups whose kernels do not share a namespace, so that a per-kernel counter sum divides into per-leg figures:
    A: santalucia_scan (k4::), least_rotation (k5::), fastq_feeder (fq::)
    B: seqhash (s2:: + k5::), fasta_feeder (fq::)
scripts/traffic_json.py turns the FETCH_SIZE / WRITE_SIZE passes of both groups into profiles/traffic.json."""
import sys

import torch

sys.path.insert(0, '.')
from poly_amd import bench_extra

CALLS = 4


def _fixed(fn, reps=0, warm=0):
    for _ in range(CALLS):
        fn()
    torch.cuda.synchronize()
    bench_extra._time.last_inner = 1
    return 1.0


_fixed.last_inner = 1
bench_extra._time = _fixed
dev = torch.device('cuda:0')
group = sys.argv[1] if len(sys.argv) > 1 else 'A'
legs = {'A': (bench_extra.tm_scan, bench_extra.rotation, bench_extra.fastq_feeder),
        'B': (bench_extra.hashing, bench_extra.fasta_feeder)}[group]
for f in legs:
    r = f(dev)
    print(f.__name__, r['workload'], flush=True)
