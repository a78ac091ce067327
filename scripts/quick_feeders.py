import sys, torch, json
sys.path.insert(0, '.')
from poly_amd import bench_extra
dev = torch.device('cuda:0')
for f in (bench_extra.fastq_feeder, bench_extra.fasta_feeder):
    r = f(dev)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != 'workload'}, flush=True)
