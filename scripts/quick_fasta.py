import sys, torch
sys.path.insert(0, '.')
from poly_amd import bench_extra
dev = torch.device('cuda:0')
r = bench_extra.fasta_feeder(dev)
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != 'workload'}, flush=True)
