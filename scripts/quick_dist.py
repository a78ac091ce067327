import sys, torch
sys.path.insert(0, '.')
from poly_amd import bench_extra
dev = torch.device('cuda:0')
r = bench_extra.distance(dev)
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ('counts_ms', 'index_build_ms', 'join_only_ms', 'distance_ms')}, r['full_matrix_one_gpu']['ms'])
r = bench_extra.fastq_feeder(dev); print('fastq', round(r['ms'], 4), round(r['file_GBs']))
r = bench_extra.fasta_feeder(dev); print('fasta', round(r['ms'], 4), round(r['file_GBs']))
